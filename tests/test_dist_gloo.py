"""world_size-2 (and 3) CPU tests of the multi-GPU path: row partitioning, CSR slicing and the
all-gather of C slabs, over the gloo backend.  The local SpMM of each rank is done by the oracle
here (test infrastructure); on GPUs it is the HIP engine writing the same slab."""
import os
import socket

import numpy as np
import pytest

from util import ALPHA, BETA, random_csr


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch
    import torch.distributed as dist
    from oracle.bindings import Oracle
    from sextans_amd import dist as sxd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = Oracle()
        rs = np.random.RandomState(77)
        M, K, N = (240, 200, 16) if mode == "even" else (251, 200, 8)
        rp, ci, v = random_csr(rs, M, K, 9, long_rows=2 if mode != "even" else 0)
        B = rs.uniform(-1, 1, K * N).astype(np.float32)
        C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
        want = C0.copy()
        o.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
        ranges = sxd.partition_rows_even(M, world) if mode == "even" else sxd.partition_rows_by_nnz(rp, world)
        r0, r1 = ranges[rank]
        lrp, lci, lv = sxd.slice_csr(rp, ci, v, r0, r1)
        # local slab: an (r1-r0) x N problem on the sliced CSR, written into the full C at row r0
        Cfull = np.full(M * N, np.nan, np.float32)
        slab = np.ascontiguousarray(C0.reshape(N, M)[:, r0:r1]).reshape(-1)
        o.spmm(r1 - r0, N, K, ALPHA, lrp, lci, lv, B, BETA, slab)
        Cfull.reshape(N, M)[:, r0:r1] = slab.reshape(N, r1 - r0)
        t = torch.from_numpy(Cfull)
        sxd.all_gather_c(t, M, N, ranges, rank)
        ok = np.array_equal(t.numpy().view(np.uint32), want.view(np.uint32))
        # row-major operands: the rank's rows are one contiguous run, the exchange is in place (dist.all_gather_rows)
        Crm = torch.full((M, N), float("nan"))
        Crm[r0:r1] = torch.from_numpy(slab.reshape(N, r1 - r0).T.copy())
        sxd.all_gather_rows(Crm, ranges, rank)
        ok = ok and np.array_equal(np.ascontiguousarray(Crm.numpy().T).reshape(-1).view(np.uint32), want.view(np.uint32))
        # single-collective form: the slab is written packed (ldc_out = lmax) into the staging buffer
        sg = sxd.SlabGather(M, N, ranges, rank, torch.device("cpu"))
        sg.local_slab()[:, :r1 - r0] = torch.from_numpy(slab.reshape(N, r1 - r0))
        sg.gather()
        t2 = torch.full((M * N,), float("nan"))
        sg.unpack_into(t2)
        ok = ok and np.array_equal(t2.numpy().view(np.uint32), want.view(np.uint32))
        # pipelined form: chunked compute (oracle here) with async all-gathers in flight; nnz-balanced (unequal)
        # ranges pad every chunk to the longest chunk of any rank, and `align` snaps this rank's cut positions
        # (the engine snaps them to its kernels' row-block boundaries) -- exchanged between ranks at set-up
        align = None if mode == "even" else (lambda r: r // 7 * 7)
        pg = sxd.PipelinedSlabGather(M, N, ranges, rank, torch.device("cpu"), nchunks=3, align=align)
        keep = []

        def chunk(c0, c1, out_ptr, ld_out, first):
            part = np.ascontiguousarray(C0.reshape(N, M)[:, r0 + c0:r0 + c1]).reshape(-1)
            crp, cci, cv = sxd.slice_csr(rp, ci, v, r0 + c0, r0 + c1)
            o.spmm(c1 - c0, N, K, ALPHA, crp, cci, cv, B, BETA, part)
            idx = [i for i, ch in enumerate(pg.chunks) if ch == (c0, c1)][0]
            assert pg.S[idx][rank].data_ptr() == out_ptr and ld_out == pg.lmax[idx] >= c1 - c0
            assert align is None or c0 % 7 == 0
            pg.S[idx][rank][:, :c1 - c0].copy_(torch.from_numpy(part.reshape(N, c1 - c0)))
            keep.append(first)
        pg.run(chunk)
        t3 = torch.full((M * N,), float("nan"))
        pg.finish(t3)
        ok = ok and keep[0] is True and not any(keep[1:]) and np.array_equal(t3.numpy().view(np.uint32), want.view(np.uint32))
        assert pg.even == (mode == "even")
        # every rank derives the hub-split threshold from the non-zeros of the WHOLE matrix (engine option "global_nnz")
        ok = ok and sxd.global_nnz(int(lrp[-1])) == int(rp[-1])
        q.put((rank, ok, ranges))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "even"), (2, "nnz"), (3, "nnz")])
def test_row_partition_and_allgather(world, mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


def test_partition_properties():
    from sextans_amd import dist as sxd
    rs = np.random.RandomState(1)
    rp, _, _ = random_csr(rs, 1000, 500, 20, long_rows=3)
    for world in (1, 2, 4, 8):
        rg = sxd.partition_rows_by_nnz(rp, world)
        assert rg[0][0] == 0 and rg[-1][1] == 1000 and all(rg[i][1] == rg[i + 1][0] for i in range(world - 1))
        loads = [int(rp[b] - rp[a]) for a, b in rg]
        assert max(loads) - min(loads) <= 2 * int(np.diff(rp).max())
        ev = sxd.partition_rows_even(1000, world)
        assert ev[0][0] == 0 and ev[-1][1] == 1000 and len({b - a for a, b in ev}) == 1
    # the C-ABI twin gives the same cuts
    from sextans_amd import api
    for world in (1, 2, 3, 8):
        assert api.partition_rows_by_nnz(rp, world) == [(int(a), int(b)) for a, b in sxd.partition_rows_by_nnz(rp, world)]
    assert api.partition_rows_by_nnz(np.array([0, 5, 5], np.int32), 4) == sxd.partition_rows_by_nnz(np.array([0, 5, 5]), 4)
    # degenerate: more ranks than rows, empty matrix
    assert sxd.partition_rows_by_nnz(np.array([0, 5, 5]), 4)[-1][1] == 2
    assert sxd.partition_rows_by_nnz(np.zeros(4, np.int32), 2)[-1][1] == 3
    lrp, lci, lv = sxd.slice_csr(rp, np.arange(rp[-1]), np.arange(rp[-1]), 10, 20)
    assert lrp[0] == 0 and lrp[-1] == len(lci) == rp[20] - rp[10]


def _balance_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch
    import torch.distributed as dist
    from sextans_amd import dist as sxd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        M = 1003                                         # not a multiple of the world size: the even slices differ in length
        rp, _, _ = random_csr(np.random.RandomState(5), M, 300, 12, long_rows=4)
        e0, e1 = sxd.partition_rows_even(M, world)[rank]
        local = torch.from_numpy((rp[e0:e1 + 1] - rp[e0]).astype(np.int32))   # what a rank holds: ITS slice, rebased
        got = sxd.balanced_ranges_from_even_slices(local, M, rank)
        ok = got == sxd.partition_rows_by_nnz(rp, world)
        try:                                             # a slice of the wrong length is refused, not gathered
            sxd.balanced_ranges_from_even_slices(local[:-1], M, rank)
            ok = False
        except ValueError:
            pass
        q.put((rank, ok, got))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_nnz_balanced_ranges_without_a_whole_row_ptr(world):
    """bench.py's N > 1 default: no rank holds the matrix; the cuts come out of all-gathered row lengths and equal the cuts
    of the whole row_ptr on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_balance_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert len({tuple(r) for _, _, r in res}) == 1
