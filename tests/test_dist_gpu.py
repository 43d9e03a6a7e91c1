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
        for k, val in dict(kernel=0, split_rows=0, bucket_rows=0).items():
            engine.set_option(k, val)
        engine.set_matrix_csr(M, K, rp, ci, v)
        dB = torch.from_numpy(B).cuda(); dCin = torch.from_numpy(C0).cuda(); dC = torch.zeros(M * N, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        engine.spmm_device(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr(), dC.data_ptr(), M, st)
        for ranges in ([(0, M)], ):
            sxd.all_gather_c(dC, M, N, ranges, 0, _force=True)          # even: coalesced in-place columns
        torch.cuda.synchronize()
        assert np.array_equal(dC.cpu().numpy().view(np.uint32), want.view(np.uint32))
        # single-collective form used by bench.py at N > 1: packed slab (ldc_out = lmax) written straight
        # into the staging buffer by the kernel, one all_gather_into_tensor, strided unpack
        for ranges in ([(0, M)], ):
            sg = sxd.SlabGather(M, N, ranges, 0, torch.device("cuda", 0))
            engine.spmm_device2(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr(), M, sg.local_ptr(), sg.lmax, st)
            sg.gather(_force=True)
            out = torch.full((M * N,), float("nan"), device="cuda")
            sg.unpack_into(out)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
        # pipelined form (bench.py at N > 1): row chunks via sextans_spmm_device_rows, B panels repacked once,
        # async all-gathers overlapping the next chunk
        pg = sxd.PipelinedSlabGather(M, N, [(0, M)], 0, torch.device("cuda", 0), nchunks=4)

        def chunk(c0, c1, out_ptr, ld_out, first):
            engine.spmm_device_rows(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr() + 4 * c0, M, out_ptr, ld_out,
                                    c0, c1, reuse_b_panels=not first, stream=st)
        pg.run(chunk, _force=True)
        out = torch.full((M * N,), float("nan"), device="cuda")
        pg.finish(out)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
        # a slab of rows [r0, r1) of a taller C_in (ldc_in = M) into a packed slab (ldc_out = r1 - r0)
        r0, r1 = 100, 420
        lrp, lci, lv = sxd.slice_csr(rp, ci, v, r0, r1)
        engine.set_matrix_csr(r1 - r0, K, lrp, lci, lv)
        slab = torch.full(((r1 - r0) * N,), float("nan"), device="cuda")
        engine.spmm_device2(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr() + 4 * r0, M, slab.data_ptr(), r1 - r0, st)
        torch.cuda.synchronize()
        got = slab.cpu().numpy().reshape(N, r1 - r0)
        assert np.array_equal(got.view(np.uint32), want.reshape(N, M)[:, r0:r1].copy().view(np.uint32))
    finally:
        dist.destroy_process_group()


def test_native_dist_spmm_single_rank(engine, oracle, sx):
    """sextans_dist_spmm (RCCL called from the C ABI, no torch.distributed) on a 1-rank communicator: chunked
    slab, ncclAllGather on the engine's communication stream, unpack -- and the LDS-panel kernel on
    block-aligned row chunks (FEM matrix)."""
    import torch
    from sextans_amd import api
    comm = api.dist_comm_init(0, 1, 0, api.dist_unique_id())
    try:
        st = torch.cuda.current_stream().cuda_stream
        rs = np.random.RandomState(4)
        for name in ("random", "fem"):
            if name == "random":
                M, K, N = 2000, 1500, 16
                rp, ci, v = random_csr(rs, M, K, 9, long_rows=1)
            else:
                rp, ci, v = api.gen_fem3d_host(12, 11, 10, 3, 7)
                M = K = 12 * 11 * 10 * 3
                N = 24
            B = rs.uniform(-1, 1, K * N).astype(np.float32)
            C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
            want = C0.copy()
            oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
            for k, val in dict(kernel=0, lanes_per_row=4, exact=1, split_rows=0, bucket_rows=0).items():
                engine.set_option(k, val)
            engine.set_matrix_csr(M, K, rp, ci, v)
            dB = torch.from_numpy(B).cuda(); dCin = torch.from_numpy(C0).cuda()
            for nchunks in (1, 3):
                out = torch.full((M * N,), float("nan"), device="cuda")
                engine.dist_spmm(comm, 1, 0, [(0, M)], N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr(), M, out.data_ptr(), M,
                                 nchunks=nchunks, stream=st)
                torch.cuda.synchronize()
                assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32)), (name, nchunks)
                if name == "fem":          # chunk cuts are snapped to row-block boundaries: every chunk keeps the panel kernel
                    assert engine.last_kernel() in ("spmm_csr_panel", "spmm_csr_panel_v2"), nchunks
            if name == "fem":
                # row-range calls cut at sextans_align_row boundaries keep the LDS-panel kernel
                cuts = [0, engine.align_row(N, M // 3), engine.align_row(N, 2 * M // 3), M]
                assert cuts[1] > 0 and cuts[2] > cuts[1]
                out = torch.full((M * N,), float("nan"), device="cuda")
                for i in range(3):
                    c0, c1 = cuts[i], cuts[i + 1]
                    slab = torch.full(((c1 - c0) * N,), float("nan"), device="cuda")
                    engine.spmm_device_rows(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr() + 4 * c0, M, slab.data_ptr(),
                                            c1 - c0, c0, c1, reuse_b_panels=i > 0, stream=st)
                    assert engine.last_kernel() in ("spmm_csr_panel", "spmm_csr_panel_v2")
                    out.view(N, M)[:, c0:c1] = slab.view(N, c1 - c0)
                torch.cuda.synchronize()
                assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
                # an unaligned cut falls back to the gather kernel, same bits
                slab = torch.full(((M - 5) * N,), float("nan"), device="cuda")
                engine.spmm_device_rows(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr() + 4 * 5, M, slab.data_ptr(), M - 5, 5, M,
                                        stream=st)
                torch.cuda.synchronize()
                assert engine.last_kernel() == "spmm_csr_rowgroup"
                assert np.array_equal(slab.cpu().numpy().reshape(N, M - 5).view(np.uint32),
                                      want.reshape(N, M)[:, 5:].copy().view(np.uint32))
    finally:
        api.dist_comm_destroy(comm)


@pytest.mark.parametrize("exchange", ["allgather", "broadcast_runs"])
def test_native_dist_spmm_rowmajor_single_rank(engine, oracle, sx, exchange, monkeypatch):
    """sextans_dist_spmm_rm on a 1-rank communicator: the slab computed in place inside row-major C_out by the row-major entry point,
    exchanged in place (ncclAllGather, or the group of ncclBroadcast that ranges of unequal length use), ldc == N and ldc > N (packed
    copy), natural / brick / gather-kernel matrices; C_in == C_out allowed."""
    import torch
    from sextans_amd import api
    if exchange == "broadcast_runs":   # (a measurement switch: the grouped-broadcast exchange on ranges of equal length too)
        monkeypatch.setenv("SEXTANS_DEBUG_OPTIONS", "1")
        engine.set_option("dist_broadcast_runs", 1)
    comm = api.dist_comm_init(0, 1, 0, api.dist_unique_id())
    try:
        st = torch.cuda.current_stream().cuda_stream
        rs = np.random.RandomState(6)
        for name in ("random", "fem", "fem bricks"):
            if name == "random":
                M, K, N = 2003, 1500, 16
                rp, ci, v = random_csr(rs, M, K, 9)
            elif name == "fem":
                rp, ci, v = api.gen_fem3d_host(12, 11, 10, 3, 7); M = K = 12 * 11 * 10 * 3; N = 24
            else:
                rp, ci, v = api.gen_fem3d_host(20, 19, 18, 3, 5); M = K = 20 * 19 * 18 * 3; N = 32
            B = rs.uniform(-1, 1, (K, N)).astype(np.float32); C0 = rs.uniform(-1, 1, (M, N)).astype(np.float32)
            w = np.ascontiguousarray(C0.T).reshape(-1).copy()
            oracle.spmm(M, N, K, ALPHA, rp, ci, v, np.ascontiguousarray(B.T).reshape(-1), BETA, w)
            want = np.ascontiguousarray(w.reshape(N, M).T)
            engine.set_matrix_csr(M, K, rp, ci, v)
            dB = torch.from_numpy(B).cuda()
            for ld in (N, N + 8):
                cin = torch.full((M, ld), 3.0, device="cuda"); cin[:, :N] = torch.from_numpy(C0).cuda()
                out = torch.full((M, ld), -5.0, device="cuda")
                engine.dist_spmm_rm(comm, 1, 0, [(0, M)], N, ALPHA, dB.data_ptr(), N, BETA, cin.data_ptr(), ld, out.data_ptr(), ld, stream=st)
                torch.cuda.synchronize()
                got = out.cpu().numpy()
                assert np.array_equal(np.ascontiguousarray(got[:, :N]).view(np.uint32), want.view(np.uint32)), (name, ld, engine.last_kernel())
                assert np.all(got[:, N:] == -5.0)
                assert "rowmajor" in engine.last_kernel(), (name, engine.last_kernel())
                engine.dist_spmm_rm(comm, 1, 0, [(0, M)], N, ALPHA, dB.data_ptr(), N, BETA, cin.data_ptr(), ld, cin.data_ptr(), ld, stream=st)   # in place
                torch.cuda.synchronize()
                assert np.array_equal(np.ascontiguousarray(cin.cpu().numpy()[:, :N]).view(np.uint32), want.view(np.uint32)), (name, ld)
        with pytest.raises(Exception):
            engine.dist_spmm_rm(comm, 1, 0, [(0, M - 1)], N, ALPHA, dB.data_ptr(), N, BETA, cin.data_ptr(), ld, out.data_ptr(), ld, stream=st)
    finally:
        api.dist_comm_destroy(comm)
        if exchange == "broadcast_runs":
            engine.set_option("dist_broadcast_runs", 0)


@pytest.mark.parametrize("with_comm", [True, False])
def test_clustered_order_chunks_keep_the_reordered_form(sx, oracle, with_comm):
    """VERDICT r04 task 4a: a slab that runs on a graph-clustered plan (a mesh in a random node order) used to fall back to the
    natural-order forms as soon as sextans_dist_spmm cut it into chunks.  Chunks are now ranges of the plan's row blocks: every chunk
    runs the reordered form, the slabs travel in clustered order and are unpacked through the position -> row tables.  1-rank RCCL
    communicator (and no communicator at all): bit-identical to cpu_spmm_CSR for 1, 2, 4 and 7 chunks."""
    import torch
    from sextans_amd import api, meshgen
    rp, ci, v = api.gen_fem3d_host(30, 28, 26, 3, 7)
    M = K = 30 * 28 * 26 * 3
    rp, ci, v = meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M // 3, 3, 9))
    rs = np.random.RandomState(2)
    comm = api.dist_comm_init(0, 1, 0, api.dist_unique_id()) if with_comm else None
    st = torch.cuda.current_stream().cuda_stream
    try:
        with api.Engine(0) as e:
            e.set_matrix_csr(M, K, rp, ci, v)
            for N in (16, 48, 24):
                B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
                want = C0.copy()
                oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
                dB = torch.from_numpy(B).cuda(); dCin = torch.from_numpy(C0).cuda()
                for nchunks in (1, 2, 4, 7):
                    for rep in range(2):                     # (the second call reuses the exchanged cuts and tables)
                        out = torch.full((M * N,), float("nan"), device="cuda")
                        e.dist_spmm(comm, 1, 0, [(0, M)], N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr(), M, out.data_ptr(), M, nchunks=nchunks, stream=st)
                        torch.cuda.synchronize()
                        assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32)), (N, nchunks, rep, e.last_kernel())
                    assert int(e.get_stat("row_cluster")) == 2
                    if N % 16 == 0:
                        assert e.last_kernel() == "spmm_csr_panel_v2_reordered", (N, nchunks, e.last_kernel())
                # a whole-matrix call in between (other B panels, other staging) does not disturb the next chunked call
                out = torch.empty(M * N, device="cuda")
                e.spmm_device(N, float(ALPHA), dB.data_ptr(), K, float(BETA), dCin.data_ptr(), out.data_ptr(), M, st)
                torch.cuda.synchronize()
                assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
    finally:
        if comm is not None:
            api.dist_comm_destroy(comm)


def _rccl_worker(rank, world, port, q):
    """One rank of a real multi-GPU run: its own GPU, its own engine on its row range of A, B replicated,
    RCCL all-gather of the C slabs (single-collective and pipelined forms)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch
    import torch.distributed as dist
    from oracle.bindings import Oracle
    from sextans_amd import api, dist as sxd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ok = {}
    try:
        o = Oracle()
        rs = np.random.RandomState(123)
        M, K, N = 6400, 5000, 16
        rp, ci, v = random_csr(rs, M, K, 12, long_rows=1)
        B = rs.uniform(-1, 1, K * N).astype(np.float32)
        C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
        want = C0.copy()
        o.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
        st = torch.cuda.current_stream().cuda_stream
        dB = torch.from_numpy(B).to(dev); dCin = torch.from_numpy(C0).to(dev)
        with api.Engine(rank) as e:
            for mode in ("even", "nnz"):
                ranges = sxd.partition_rows_even(M, world) if mode == "even" else sxd.partition_rows_by_nnz(rp, world)
                r0, r1 = ranges[rank]
                lrp, lci, lv = sxd.slice_csr(rp, ci, v, r0, r1)
                e.set_matrix_csr(r1 - r0, K, lrp, lci, lv)
                sg = sxd.SlabGather(M, N, ranges, rank, dev)
                e.spmm_device2(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr() + 4 * r0, M, sg.local_ptr(), sg.lmax, st)
                sg.gather()
                out = torch.full((M * N,), float("nan"), device=dev)
                sg.unpack_into(out)
                torch.cuda.synchronize()
                ok["slab_" + mode] = bool(np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32)))
                pg = sxd.PipelinedSlabGather(M, N, ranges, rank, dev, nchunks=3)

                def chunk(c0, c1, out_ptr, ld_out, first):
                    e.spmm_device_rows(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr() + 4 * (r0 + c0), M, out_ptr,
                                       ld_out, c0, c1, reuse_b_panels=not first, stream=st)
                pg.run(chunk)
                out = torch.full((M * N,), float("nan"), device=dev)
                pg.finish(out)
                torch.cuda.synchronize()
                ok["pipelined_" + mode] = bool(np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32)))
                # the native form: RCCL from the C ABI; the id travels through the torch store
                ids = [api.dist_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                comm = api.dist_comm_init(rank, world, rank, ids[0])
                out = torch.full((M * N,), float("nan"), device=dev)
                if mode == "nnz":   # (round 6: the collective preparation over real RCCL; the even-rows leg keeps the lazy path)
                    e.dist_prepare(comm, world, rank, ranges, N, nchunks=3, form=0, stream=st)
                    x0 = e.get_stat("dist_setup_exchanges")
                e.dist_spmm(comm, world, rank, ranges, N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr(), M, out.data_ptr(), M,
                            nchunks=3, stream=st)
                torch.cuda.synchronize()
                ok["native_" + mode] = bool(np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32)))
                if mode == "nnz":
                    ok["prepared_call_exchanged_nothing"] = e.get_stat("dist_setup_exchanges") == x0
                # row-major operands: slabs written in place, in-place all-gather (equal ranges) / grouped broadcasts (nnz-balanced)
                want_rm = np.ascontiguousarray(want.reshape(N, M).T)
                dBr = torch.from_numpy(np.ascontiguousarray(B.reshape(N, K).T)).to(dev)
                for ld in (N, N + 4):
                    cin = torch.zeros((M, ld), device=dev); cin[:, :N] = torch.from_numpy(np.ascontiguousarray(C0.reshape(N, M).T)).to(dev)
                    out = torch.full((M, ld), float("nan"), device=dev)
                    e.dist_spmm_rm(comm, world, rank, ranges, N, ALPHA, dBr.data_ptr(), N, BETA, cin.data_ptr(), ld, out.data_ptr(), ld, stream=st)
                    torch.cuda.synchronize()
                    ok[f"native_rowmajor_{mode}_ld{ld}"] = bool(np.array_equal(np.ascontiguousarray(out.cpu().numpy()[:, :N]).view(np.uint32), want_rm.view(np.uint32)))
                if mode == "even":   # blocked-ELL bf16 over the same communicator: block-row ranges, packed slabs, one all-gather
                    Mb, Kb, Nb, Wb = 1024, 1024, 64, 5
                    bcol, bval = api.gen_bell_host(Mb, Kb, Wb, 9)
                    B16 = api.gen_uniform_bf16_host(Kb * Nb, 3)
                    Cb = np.random.RandomState(1).uniform(-1, 1, Mb * Nb).astype(np.float32)
                    dBb = torch.from_numpy(B16.view(np.int16)).to(dev); dCb = torch.from_numpy(Cb).to(dev)
                    with api.Engine(rank) as eb:
                        eb.set_matrix_bell(Mb, Kb, Wb, bcol, bval)
                        whole = torch.zeros(Mb * Nb, device=dev)
                        eb.spmm_bell_device(Nb, ALPHA, dBb.data_ptr(), Kb, BETA, dCb.data_ptr(), whole.data_ptr(), Mb, st)
                        rg = [(g * (Mb // world // 32) * 32, (g + 1) * (Mb // world // 32) * 32 if g + 1 < world else Mb) for g in range(world)]
                        b0, b1 = rg[rank]
                        eb.set_matrix_bell(b1 - b0, Kb, Wb, bcol[b0 // 32 * Wb:b1 // 32 * Wb], bval[b0 // 32 * Wb * 1024:b1 // 32 * Wb * 1024])
                        outb = torch.full((Mb * Nb,), float("nan"), device=dev)
                        eb.dist_spmm_bell(comm, world, rank, rg, Nb, ALPHA, dBb.data_ptr(), Kb, BETA, dCb.data_ptr(), Mb, outb.data_ptr(), Mb, stream=st)
                        torch.cuda.synchronize()
                        ok["native_blocked_ell"] = bool(torch.equal(outb, whole))
                api.dist_comm_destroy(comm)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two MI355X on one node (runs as soon as the box has them)")
def test_rccl_two_ranks():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok in res:
        assert ok and all(ok.values()), (rank, ok)


def test_cpp_example_multi_gpu_through_the_c_abi(sx):
    """examples/dist_spmm.cpp: one host thread per GPU, RCCL bound from the C ABI, no Python in the data path.  Runs on
    however many gfx950 devices the box has (1 here; the same binary shards over 2-8)."""
    import subprocess
    exe = os.path.join(os.path.dirname(sx.api.CLI_PATH), "dist_spmm")
    assert os.path.exists(exe)
    nasa = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matrices", "nasa4704", "nasa4704.mtx")
    for n in ("16", "40"):
        r = subprocess.run([exe, nasa, n], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "all ranks match the single-GPU result" in r.stdout, r.stdout + r.stderr
    # ... and with the row-major form next to it (sextans_dist_spmm_rm: slabs in place, in-place exchange)
    r = subprocess.run([exe, nasa, "24", "8", "rm"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "column-major and row-major forms: all ranks match the single-GPU result" in r.stdout, r.stdout + r.stderr


def test_new_b_after_a_fused_chunk_repacks_for_later_chunks(engine, oracle):
    """Round-2 ADVICE: a row-range call that stages straight from column-major B (no repack) left the panel
    workspace marked as valid; the next pipelined SpMM with ANOTHER B then reused the first B's panels in a later chunk
    that needs them (here: the chunk holding a long row, which goes through the piece kernel and its panels)."""
    import torch
    from sextans_amd import api
    rp, ci, v = api.gen_fem3d_host(12, 11, 10, 3, 7)
    M = K = 12 * 11 * 10 * 3
    N = 16
    rs = np.random.RandomState(11)
    # one long row in the last third: replace row `hub` by 400 random distinct columns
    hub = M - 500
    cols = np.sort(rs.choice(K, 400, replace=False)).astype(np.int32)
    a, b = int(rp[hub]), int(rp[hub + 1])
    ci = np.concatenate([ci[:a], cols, ci[b:]]).astype(np.int32)
    v = np.concatenate([v[:a], rs.uniform(-1, 1, 400).astype(np.float32), v[b:]]).astype(np.float32)
    rp = rp.copy(); rp[hub + 1:] += 400 - (b - a)
    for k, val in dict(kernel=0, lanes_per_row=4, exact=1, split_rows=0, bucket_rows=200, fuse_b=1).items():
        engine.set_option(k, val)
    try:
        engine.set_matrix_csr(M, K, rp, ci, v)
        assert engine.get_stat("piece_path_rows") == 1
        cut = engine.align_row(N, M // 2)
        assert 0 < cut < hub
        st = torch.cuda.current_stream().cuda_stream
        C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
        dCin = torch.from_numpy(C0).cuda()
        for trial in range(3):                      # a different B every time, same engine, same workspace
            B = rs.uniform(-1, 1, K * N).astype(np.float32)
            want = C0.copy()
            oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
            dB = torch.from_numpy(B).cuda()
            out = torch.full((M * N,), float("nan"), device="cuda")
            kernels = []
            for i, (c0, c1) in enumerate(((0, cut), (cut, M))):
                engine.spmm_device_rows(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr() + 4 * c0, M,
                                        out.data_ptr() + 4 * c0, M, c0, c1, reuse_b_panels=i > 0, stream=st)
                kernels.append(engine.last_kernel())
            torch.cuda.synchronize()
            # chunk 1: column-major staging (no repack); chunk 2: long row => repacked panels (register-resident form + piece kernel)
            assert kernels == ["spmm_csr_panel", "spmm_csr_panel_v2+hub_pieces"], kernels
            assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32)), trial
    finally:
        for k, val in dict(bucket_rows=-1, split_rows=0).items():
            engine.set_option(k, val)


def test_hub_split_threshold_follows_the_global_nnz(engine, sx):
    """VERDICT r02 task 6c: with "split_rows" = -1 the threshold T = max(1024, nnz / 16384) used the rank-LOCAL nnz, so a
    row-partitioned SpMM could cut a hub row differently from a single GPU.  With option "global_nnz" (set by
    sextans_dist_spmm from an exchanged sum, by bench.py from dist.global_nnz) both halves of a power-law matrix on two
    engines give the single-engine result bit for bit and the same list of re-associated rows."""
    import torch
    from sextans_amd import api, dist as sxd
    M = K = 600_000
    N = 8
    rp, ci, v = api.gen_powerlaw_host(M, K, 6, 120, 400_000, 7)
    nnz = int(rp[-1])
    T = nnz // 16384
    assert T > 1100, nnz                       # the global threshold is above the floor of 1024 ...
    lens = np.diff(rp)
    half = sxd.partition_rows_by_nnz(rp, 2)
    between = np.nonzero((lens > 1024) & (lens <= T))[0]
    assert len(between) > 0                    # ... and some rows sit between the local and the global one
    st = torch.cuda.current_stream().cuda_stream
    rs = np.random.RandomState(5)
    dB = torch.from_numpy(rs.uniform(-1, 1, K * N).astype(np.float32)).cuda()
    dCin = torch.from_numpy(rs.uniform(-1, 1, M * N).astype(np.float32)).cuda()
    opts = dict(kernel=0, lanes_per_row=0, exact=1, split_rows=-1, bucket_rows=-1, global_nnz=0)
    try:
        for k, val in opts.items():
            engine.set_option(k, val)
        engine.set_matrix_csr(M, K, rp, ci, v)
        whole = torch.empty(M * N, device="cuda")
        engine.spmm_device(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr(), whole.data_ptr(), M, st)
        torch.cuda.synchronize()
        hubs = engine.reassociated_rows()
        assert engine.get_stat("split_threshold") == T and np.array_equal(hubs, np.nonzero(lens > T)[0])
        for use_global in (True, False):
            parts = torch.full((M * N,), float("nan"), device="cuda")
            got_hubs = []
            for r0, r1 in half:
                lrp, lci, lv = sxd.slice_csr(rp, ci, v, r0, r1)
                with sx.Engine(0) as e2:
                    for k, val in opts.items():
                        e2.set_option(k, val)
                    if use_global:
                        e2.set_option("global_nnz", nnz)
                    e2.set_matrix_csr(r1 - r0, K, lrp, lci, lv)
                    e2.spmm_device2(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr() + 4 * r0, M,
                                    parts.data_ptr() + 4 * r0, M, st)
                    torch.cuda.synchronize()
                    got_hubs.append(e2.reassociated_rows() + r0)
            same = bool(torch.equal(parts, whole))
            if use_global:
                assert same and np.array_equal(np.concatenate(got_hubs), hubs)
            else:   # the defect being fixed: local thresholds (1024 here) re-associate more rows than one GPU does
                assert len(np.concatenate(got_hubs)) > len(hubs)
    finally:
        engine.set_matrix_csr(1, 1, np.array([0, 0], np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32))
        for k, val in dict(split_rows=0, bucket_rows=-1, global_nnz=0).items():
            engine.set_option(k, val)
