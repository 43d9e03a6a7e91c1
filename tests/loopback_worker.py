"""Worker of tests/test_dist_loopback_gpu.py: runs the NATIVE multi-GPU entry points (sextans_dist_spmm, _rm, _bell, sextans_dist_prepare)
with world > 1 on ONE GPU -- every rank is a host thread with its own engine, stream, B and C, and the collectives come from the
loopback communicator tests/fake_rccl.cpp bound through sextans_dist_bind_library.  Every rank's whole C must be bit-identical to the
CPU oracle (cpu_spmm_CSR restated, oracle/sextans_oracle.c).

    python tests/loopback_worker.py <scenario> <world>

A separate process per scenario: a rank that fails in front of a collective leaves its peers in a host barrier for ever, and only a
process can be killed (the test's timeout does that).  Prints "LOOPBACK OK ..." on success, exits non-zero otherwise."""
import os
import sys
import threading
import traceback

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from util import ALPHA, BETA, random_csr  # noqa: E402


def fake_rccl_path():
    """Builds tests/_build/libfake_rccl.so (hipcc, host API only) when missing or stale."""
    import shutil
    import subprocess
    src = os.path.join(HERE, "fake_rccl.cpp")
    out = os.path.join(HERE, "_build", "libfake_rccl.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        subprocess.run([hipcc, "-x", "hip", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread",
                        "-fgpu-default-stream=per-thread", "-o", out + ".tmp", src], check=True)
        os.replace(out + ".tmp", out)
    return out


class Ranks:
    """world host threads = world ranks on device 0."""

    def __init__(self, world):
        from sextans_amd import api
        self.world = world
        self.api = api
        self.uid = api.dist_unique_id()
        self.errors = [None] * world
        self.results = [None] * world

    def run(self, fn, timeout=900):
        import torch

        def body(rank):
            try:
                torch.cuda.set_device(0)
                st = torch.cuda.Stream()
                comm = self.api.dist_comm_init(0, self.world, rank, self.uid)
                try:
                    with torch.cuda.stream(st):
                        self.results[rank] = fn(rank, comm, st.cuda_stream)
                    st.synchronize()
                finally:
                    self.api.dist_comm_destroy(comm)
            except BaseException:   # noqa: BLE001  (reported by the main thread)
                self.errors[rank] = traceback.format_exc()

        ts = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(self.world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout)
        hung = [r for r, t in enumerate(ts) if t.is_alive()]
        bad = [(r, e) for r, e in enumerate(self.errors) if e]
        if hung or bad:
            for r, e in bad:
                print(f"--- rank {r} failed:\n{e}", file=sys.stderr)
            raise SystemExit(f"loopback: ranks hung {hung}, ranks failed {[r for r, _ in bad]}")
        return self.results


def ranges_for(mode, rp, M, world):
    from sextans_amd import dist as sxd
    return sxd.partition_rows_even(M, world) if mode == "even" else sxd.partition_rows_by_nnz(rp, world)


def matrices(which):
    """(name, rp, ci, v, M, K, N list)"""
    from sextans_amd import api, meshgen
    rs = np.random.RandomState(17)
    if which == "random":
        M, K = 6403, 5000
        rp, ci, v = random_csr(rs, M, K, 12, long_rows=2)
        return "random", rp, ci, v, M, K
    if which == "fem":
        rp, ci, v = api.gen_fem3d_host(14, 13, 12, 3, 7)
        M = K = 14 * 13 * 12 * 3
        return "fem", rp, ci, v, M, K
    if which == "bricks":
        rp, ci, v = api.gen_fem3d_host(24, 22, 20, 3, 5)
        M = K = 24 * 22 * 20 * 3
        return "bricks", rp, ci, v, M, K
    if which == "mesh_random_order":   # a mesh in a random node order: every slab runs on its graph-clustered plan
        rp, ci, v = api.gen_fem3d_host(40, 40, 40, 3, 7)
        M = K = 40 * 40 * 40 * 3
        rp, ci, v = meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M // 3, 3, 9))
        return "mesh_random_order", rp, ci, v, M, K
    raise ValueError(which)


def scenario_colmajor(world, clustered=False):
    """sextans_dist_spmm: chunk pipeline, 1 and 4 chunks, even and nnz-balanced ranges, with and without sextans_dist_prepare."""
    import torch
    from oracle.bindings import Oracle
    from sextans_amd import api, dist as sxd
    o = Oracle()
    rs = np.random.RandomState(5)
    checked = 0
    kernels = set()
    for which in (("mesh_random_order",) if clustered else ("random", "fem")):
        name, rp, ci, v, M, K = matrices(which)
        for N in ((16, 48) if clustered else (16, 24)):
            B = rs.uniform(-1, 1, K * N).astype(np.float32)
            C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
            want = C0.copy()
            o.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
            for mode in ("even", "nnz"):
                ranges = ranges_for(mode, rp, M, world)

                def rank_fn(rank, comm, st):
                    r0, r1 = ranges[rank]
                    lrp, lci, lv = sxd.slice_csr(rp, ci, v, r0, r1)
                    dB = torch.from_numpy(B).cuda(); dCin = torch.from_numpy(C0).cuda()
                    out_ok = []
                    with api.Engine(0) as e:
                        e.set_matrix_csr(r1 - r0, K, lrp, lci, lv)
                        for nchunks in ((1, 4) if not clustered else (4, 1, 7)):
                            for prepared in (False, True):   # (lazily inside the first call, then once more through sextans_dist_prepare)
                                if prepared:
                                    e.dist_prepare(comm, world, rank, ranges, N, nchunks=nchunks, form=0, stream=st)
                                    x0 = e.get_stat("dist_setup_exchanges")
                                for rep in range(2):
                                    out = torch.full((M * N,), float("nan"), device="cuda")
                                    e.dist_spmm(comm, world, rank, ranges, N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr(), M, out.data_ptr(), M,
                                                nchunks=nchunks, stream=st)
                                    torch.cuda.current_stream().synchronize()
                                    same = np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
                                    out_ok.append((nchunks, prepared, rep, same, e.last_kernel()))
                                if prepared:   # nothing was exchanged or synchronised inside the calls that followed the preparation
                                    assert e.get_stat("dist_setup_exchanges") == x0, (rank, nchunks, x0, e.get_stat("dist_setup_exchanges"))
                    return out_ok

                for rank, res in enumerate(Ranks(world).run(rank_fn)):
                    for nchunks, prepared, rep, same, kern in res:
                        assert same, (name, N, mode, "rank", rank, "chunks", nchunks, "prepared", prepared, "rep", rep, kern)
                        kernels.add(kern)
                        checked += 1
    if clustered:
        assert "spmm_csr_panel_v2_reordered" in kernels, kernels   # clustered-order chunks ran on every rank (all ranks agreed)
    return f"{checked} rank results bit-identical; kernels {sorted(kernels)}"


def scenario_rowmajor(world):
    """sextans_dist_spmm_rm: slabs in place; equal ranges -> in-place ncclAllGather, nnz-balanced -> grouped ncclBroadcast; ldc == N, > N; in place."""
    import torch
    from oracle.bindings import Oracle
    from sextans_amd import api, dist as sxd
    o = Oracle()
    rs = np.random.RandomState(6)
    checked = 0
    kernels = set()
    for which in ("random", "fem", "bricks"):
        name, rp, ci, v, M, K = matrices(which)
        if which == "random":   # (a multiple of the world size so that "even" really is the all-gather form)
            M = M // world * world
            rp = rp[:M + 1].copy(); ci = ci[:rp[-1]].copy(); v = v[:rp[-1]].copy()
        N = {"random": 16, "fem": 24, "bricks": 32}[which]
        B = rs.uniform(-1, 1, (K, N)).astype(np.float32); C0 = rs.uniform(-1, 1, (M, N)).astype(np.float32)
        w = np.ascontiguousarray(C0.T).reshape(-1).copy()
        o.spmm(M, N, K, ALPHA, rp, ci, v, np.ascontiguousarray(B.T).reshape(-1), BETA, w)
        want = np.ascontiguousarray(w.reshape(N, M).T)
        for mode in ("even", "nnz"):
            ranges = ranges_for(mode, rp, M, world)

            def rank_fn(rank, comm, st):
                r0, r1 = ranges[rank]
                lrp, lci, lv = sxd.slice_csr(rp, ci, v, r0, r1)
                dB = torch.from_numpy(B).cuda()
                res = []
                with api.Engine(0) as e:
                    e.set_matrix_csr(r1 - r0, K, lrp, lci, lv)
                    for ld in (N, N + 8):
                        e.dist_prepare(comm, world, rank, ranges, N, nchunks=-1 if ld != N else 0, form=1, stream=st)
                        x0 = e.get_stat("dist_setup_exchanges")
                        cin = torch.full((M, ld), 3.0, device="cuda"); cin[:, :N] = torch.from_numpy(C0).cuda()
                        out = torch.full((M, ld), -5.0, device="cuda")
                        e.dist_spmm_rm(comm, world, rank, ranges, N, ALPHA, dB.data_ptr(), N, BETA, cin.data_ptr(), ld, out.data_ptr(), ld, stream=st)
                        torch.cuda.current_stream().synchronize()
                        got = out.cpu().numpy()
                        same = np.array_equal(np.ascontiguousarray(got[:, :N]).view(np.uint32), want.view(np.uint32)) and bool(np.all(got[:, N:] == -5.0))
                        res.append((ld, "out of place", same, e.last_kernel()))
                        assert e.get_stat("dist_setup_exchanges") == x0
                        # in place: C_in == C_out (every rank reads only its own rows of C_in)
                        e.dist_spmm_rm(comm, world, rank, ranges, N, ALPHA, dB.data_ptr(), N, BETA, cin.data_ptr(), ld, cin.data_ptr(), ld, stream=st)
                        torch.cuda.current_stream().synchronize()
                        same = np.array_equal(np.ascontiguousarray(cin.cpu().numpy()[:, :N]).view(np.uint32), want.view(np.uint32))
                        res.append((ld, "in place", same, e.last_kernel()))
                return res

            for rank, res in enumerate(Ranks(world).run(rank_fn)):
                for ld, how, same, kern in res:
                    assert same, (name, mode, "rank", rank, "ld", ld, how, kern)
                    assert "rowmajor" in kern, kern
                    kernels.add(kern)
                    checked += 1
    return f"{checked} rank results bit-identical; kernels {sorted(kernels)}"


def scenario_bell(world):
    """sextans_dist_spmm_bell over block-row ranges (multiples of 32, unequal): bit-identical to the single-GPU call on the whole matrix."""
    import torch
    from sextans_amd import api
    Mb, Kb, Nb, Wb = 32 * 37, 1024, 64, 5
    bcol, bval = api.gen_bell_host(Mb, Kb, Wb, 9)
    B16 = api.gen_uniform_bf16_host(Kb * Nb, 3)
    Cb = np.random.RandomState(1).uniform(-1, 1, Mb * Nb).astype(np.float32)
    torch.cuda.set_device(0)
    dBb = torch.from_numpy(B16.view(np.int16)).cuda(); dCb = torch.from_numpy(Cb).cuda()
    st0 = torch.cuda.current_stream().cuda_stream
    with api.Engine(0) as eb:
        eb.set_matrix_bell(Mb, Kb, Wb, bcol, bval)
        whole = torch.zeros(Mb * Nb, device="cuda")
        eb.spmm_bell_device(Nb, ALPHA, dBb.data_ptr(), Kb, BETA, dCb.data_ptr(), whole.data_ptr(), Mb, st0)
        torch.cuda.synchronize()
    nbr = Mb // 32
    cuts = [nbr * g // world for g in range(world)] + [nbr]
    if world > 2:
        cuts[1] = max(cuts[1] - 1, cuts[0])   # unequal ranges: slabs padded to the longest
    rg = [(cuts[g] * 32, cuts[g + 1] * 32) for g in range(world)]

    def rank_fn(rank, comm, st):
        b0, b1 = rg[rank]
        with api.Engine(0) as e:
            e.set_matrix_bell(b1 - b0, Kb, Wb, bcol[b0 // 32 * Wb:b1 // 32 * Wb], bval[b0 // 32 * Wb * 1024:b1 // 32 * Wb * 1024])
            e.dist_prepare(comm, world, rank, rg, Nb, form=2, stream=st)
            x0 = e.get_stat("dist_setup_exchanges")
            ok = []
            for rep in range(2):
                out = torch.full((Mb * Nb,), float("nan"), device="cuda")
                e.dist_spmm_bell(comm, world, rank, rg, Nb, ALPHA, dBb.data_ptr(), Kb, BETA, dCb.data_ptr(), Mb, out.data_ptr(), Mb, stream=st)
                torch.cuda.current_stream().synchronize()
                ok.append(bool(torch.equal(out, whole)))
            assert e.get_stat("dist_setup_exchanges") == x0
        return ok

    res = Ranks(world).run(rank_fn)
    assert all(all(r) for r in res), res
    return f"{world} ranks x 2 calls bit-identical to the single-GPU blocked-ELL result; ranges {rg}"


def scenario_errors(world):
    """sextans_dist_prepare returns an error on EVERY rank when one rank fails (its own code there, SEXTANS_ERR_PEER elsewhere), nobody hangs,
    and the communicator stays usable afterwards."""
    import torch
    from sextans_amd import api, dist as sxd
    name, rp, ci, v, M, K = matrices("fem")
    N = 16
    ranges = ranges_for("nnz", rp, M, world)
    bad_rank = world - 1
    rs = np.random.RandomState(2)
    B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    from oracle.bindings import Oracle
    want = C0.copy()
    Oracle().spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)

    def rank_fn(rank, comm, st):
        r0, r1 = ranges[rank]
        lrp, lci, lv = sxd.slice_csr(rp, ci, v, r0, r1)
        codes = []
        with api.Engine(0) as e:
            if rank == bad_rank:   # this rank's engine holds one row too few: its preparation fails locally
                e.set_matrix_csr(r1 - r0 - 1, K, lrp[:-1], lci[:lrp[-2]], lv[:lrp[-2]])
            else:
                e.set_matrix_csr(r1 - r0, K, lrp, lci, lv)
            for form, nchunks in ((0, 4), (1, 0)):
                try:
                    e.dist_prepare(comm, world, rank, ranges, N, nchunks=nchunks, form=form, stream=st)
                    codes.append(0)
                except api.SextansError as ex:
                    codes.append(ex.code)
            # the same communicator, now with every rank in order: prepared, run, correct
            e.set_matrix_csr(r1 - r0, K, lrp, lci, lv)
            e.dist_prepare(comm, world, rank, ranges, N, nchunks=3, form=0, stream=st)
            dB = torch.from_numpy(B).cuda(); dCin = torch.from_numpy(C0).cuda()
            out = torch.full((M * N,), float("nan"), device="cuda")
            e.dist_spmm(comm, world, rank, ranges, N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr(), M, out.data_ptr(), M, nchunks=3, stream=st)
            torch.cuda.current_stream().synchronize()
            codes.append(bool(np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))))
        return codes

    res = Ranks(world).run(rank_fn)
    for rank, codes in enumerate(res):
        expect = 9 if rank == bad_rank else 13   # SEXTANS_ERR_INVALID on the rank that failed, SEXTANS_ERR_PEER on the others
        assert codes == [expect, expect, True], (rank, codes)
    return f"all {world} ranks saw the failure of rank {bad_rank} ({res}) and recovered"


def scenario_config4_full(world):
    """BASELINE config 4 AS STATED -- synthetic 4M x 4M CSR, ~1.6e8 non-zeros, N = 16, A row-split over `world` ranks (nnz-balanced ranges), B
    replicated, all-gather of C -- at FULL SIZE, the ranks as threads on one GPU: every rank's complete C (column-major entry, 4 chunks;
    row-major entry, in place) bit-identical to the single-engine result of the same matrix, which tests/test_configs_gpu.py holds to the
    oracle on every row."""
    import torch
    from sextans_amd import api, dist as sxd
    M = K = 4_000_000
    N = 16
    torch.cuda.set_device(0)
    st0 = torch.cuda.current_stream().cuda_stream
    B = torch.empty(K * N, device="cuda"); Cin = torch.empty(M * N, device="cuda")
    api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st0); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st0)
    p, i, v, nnz = api.gen_csr_device(0, M, K, 40.0, 4)
    hrp = np.empty(M + 1, np.int32)
    api.device_copy(0, hrp.ctypes.data, p, hrp.nbytes, api.COPY_D2H)
    ranges = sxd.partition_rows_by_nnz(hrp, world)
    whole = torch.empty(M * N, device="cuda"); whole_rm = torch.empty(M * N, device="cuda")
    with api.Engine(0) as e:
        e.set_matrix_csr_device(M, K, nnz, p, i, v)
        e.spmm_device(N, ALPHA, B.data_ptr(), K, BETA, Cin.data_ptr(), whole.data_ptr(), M, st0)
        e.spmm_device_rm(N, ALPHA, B.data_ptr(), N, BETA, Cin.data_ptr(), N, whole_rm.data_ptr(), N, st0)   # (the same buffers read as row-major operands)
        torch.cuda.synchronize()
    for q in (p, i, v):
        api.device_free(0, q)

    def rank_fn(rank, comm, st):
        r0, r1 = ranges[rank]
        lp, li, lv, lnnz = api.gen_csr_device(0, M, K, 40.0, 4, r0, r1)   # the counter-based generator yields any row range
        ok = {}
        try:
            with api.Engine(0) as e:
                e.set_matrix_csr_device(r1 - r0, K, lnnz, lp, li, lv)
                out = torch.full((M * N,), float("nan"), device="cuda")
                e.dist_prepare(comm, world, rank, ranges, N, nchunks=4, form=0, stream=st)
                e.dist_spmm(comm, world, rank, ranges, N, ALPHA, B.data_ptr(), K, BETA, Cin.data_ptr(), M, out.data_ptr(), M, nchunks=4, stream=st)
                torch.cuda.current_stream().synchronize()
                ok["colmajor"] = bool(torch.equal(out, whole))
                out.fill_(float("nan"))
                e.dist_prepare(comm, world, rank, ranges, N, form=1, stream=st)
                e.dist_spmm_rm(comm, world, rank, ranges, N, ALPHA, B.data_ptr(), N, BETA, Cin.data_ptr(), N, out.data_ptr(), N, stream=st)
                torch.cuda.current_stream().synchronize()
                ok["rowmajor"] = bool(torch.equal(out, whole_rm))
                ok["kernel"] = e.last_kernel()
        finally:
            for q in (lp, li, lv):
                api.device_free(0, q)
        return ok

    res = Ranks(world).run(rank_fn, timeout=1500)
    assert all(r["colmajor"] and r["rowmajor"] for r in res), res
    return f"config 4 at full size ({nnz} non-zeros) over {world} ranks, ranges {ranges[:2]} ...: every rank's C bit-identical to one engine holding every row ({res[0]['kernel']})"


def scenario_config5_full(world):
    """BASELINE config 5 at full size (blocked-ELL 1M x 1M, 32x32 bf16 blocks, 1 % block fill, N = 256) over `world` block-row ranges:
    sextans_dist_spmm_bell on every rank, complete C bit-identical to the single-engine call ("Config 5 likewise")."""
    import torch
    from sextans_amd import api
    M = K = 1_048_576
    W, N = 328, 256
    torch.cuda.set_device(0)
    st0 = torch.cuda.current_stream().cuda_stream
    dc, dv = api.gen_bell_device(0, M, K, W, 5)
    B = torch.empty(K * N, dtype=torch.int16, device="cuda"); Cin = torch.empty(M * N, device="cuda")
    api.gen_uniform_bf16_device(0, B.data_ptr(), K * N, 51, st0); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 52, st0)
    whole = torch.empty(M * N, device="cuda")
    with api.Engine(0) as e:
        e.set_matrix_bell_device(M, K, W, dc, dv)
        e.spmm_bell_device(N, ALPHA, B.data_ptr(), K, BETA, Cin.data_ptr(), whole.data_ptr(), M, st0)
        torch.cuda.synchronize()
    nbr = M // 32
    cuts = [nbr * g // world for g in range(world)] + [nbr]
    cuts[1] -= 3                                            # unequal ranges: slabs padded to the longest
    rg = [(cuts[g] * 32, cuts[g + 1] * 32) for g in range(world)]

    def rank_fn(rank, comm, st):
        b0, b1 = rg[rank]
        with api.Engine(0) as e:
            e.set_matrix_bell_device(b1 - b0, K, W, dc + (b0 // 32) * W * 4, dv + (b0 // 32) * W * 2048)
            out = torch.full((M * N,), float("nan"), device="cuda")
            e.dist_prepare(comm, world, rank, rg, N, form=2, stream=st)
            e.dist_spmm_bell(comm, world, rank, rg, N, ALPHA, B.data_ptr(), K, BETA, Cin.data_ptr(), M, out.data_ptr(), M, stream=st)
            torch.cuda.current_stream().synchronize()
            return bool(torch.equal(out, whole))

    res = Ranks(world).run(rank_fn, timeout=1500)
    for q in (dc, dv):
        api.device_free(0, q)
    assert all(res), res
    return f"config 5 at full size over {world} block-row ranges: every rank's C bit-identical to the single-engine result"


def scenario_hub_rows(world):
    """Hub rows across ranks: with "split_rows" = -1 (part of SEXTANS_MODE_FAST) the cut threshold T = max(1024, nnz / 16384) follows the non-zeros
    of the WHOLE matrix, which only the exchange of the ranks' counts can know (sextans_dist_prepare / the first dist call): every rank
    then cuts its hub rows exactly as one GPU holding all rows does, and the N-rank result equals the 1-GPU result bit for bit."""
    import torch
    from sextans_amd import api, dist as sxd
    M = K = 600_000
    N = 8
    rp, ci, v = api.gen_powerlaw_host(M, K, 6, 120, 400_000, 7)
    nnz = int(rp[-1])
    assert nnz // 16384 > 1100                              # the global threshold lies above the floor a rank's own count would give
    rs = np.random.RandomState(5)
    B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    torch.cuda.set_device(0)
    st0 = torch.cuda.current_stream().cuda_stream
    dB = torch.from_numpy(B).cuda(); dCin = torch.from_numpy(C0).cuda()
    whole = torch.empty(M * N, device="cuda")
    with api.Engine(0) as e:
        e.set_option("mode", 1)
        e.set_matrix_csr(M, K, rp, ci, v)
        e.spmm_device(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr(), whole.data_ptr(), M, st0)
        torch.cuda.synchronize()
        hubs = e.reassociated_rows()
    assert len(hubs) > 0
    ranges = sxd.partition_rows_by_nnz(rp, world)

    def rank_fn(rank, comm, st):
        r0, r1 = ranges[rank]
        lrp, lci, lv = sxd.slice_csr(rp, ci, v, r0, r1)
        with api.Engine(0) as e:
            e.set_option("mode", 1)
            e.set_matrix_csr(r1 - r0, K, lrp, lci, lv)
            res = {}
            for form in ("lazy", "prepared"):
                if form == "prepared":
                    e.dist_prepare(comm, world, rank, ranges, N, nchunks=2, form=0, stream=st)
                out = torch.full((M * N,), float("nan"), device="cuda")
                e.dist_spmm(comm, world, rank, ranges, N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr(), M, out.data_ptr(), M, nchunks=2, stream=st)
                torch.cuda.current_stream().synchronize()
                res[form] = bool(torch.equal(out, whole))
            res["threshold"] = int(e.get_stat("split_threshold"))
            res["hubs"] = (e.reassociated_rows() + r0).tolist()
        return res

    res = Ranks(world).run(rank_fn)
    assert all(r["lazy"] and r["prepared"] for r in res), res
    assert all(r["threshold"] == nnz // 16384 for r in res), [r["threshold"] for r in res]
    assert sorted(sum((r["hubs"] for r in res), [])) == list(hubs)
    return f"{len(hubs)} hub rows cut at the global threshold {nnz // 16384} on every one of {world} ranks; all ranks bit-identical to one GPU"


def scenario_empty_ranges(world):
    """More ranks than rows with non-zeros: nnz-balanced ranges leave some ranks with NO rows (and one rank's rows may all be empty).  Every
    form must still complete on every rank with the whole C."""
    import torch
    from oracle.bindings import Oracle
    from sextans_amd import api, dist as sxd
    o = Oracle()
    rs = np.random.RandomState(3)
    M, K = (37 if world < 8 else 11), 50                    # (8 ranks over 11 mostly empty rows: nnz-balanced ranges with empty members)
    rp, ci, v = random_csr(rs, M, K, 6, empty_frac=0.6)
    out_all = {}
    for N in (8, 16):
        B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
        want = C0.copy()
        o.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
        want_rm = np.ascontiguousarray(want.reshape(N, M).T)
        for mode in ("nnz", "front"):
            ranges = sxd.partition_rows_by_nnz(rp, world) if mode == "nnz" else [(0, M)] + [(M, M)] * (world - 1)   # everything on rank 0
            assert mode == "nnz" or any(a == b for a, b in ranges)

            def rank_fn(rank, comm, st):
                r0, r1 = ranges[rank]
                lrp, lci, lv = sxd.slice_csr(rp, ci, v, r0, r1)
                dB = torch.from_numpy(B).cuda(); dCin = torch.from_numpy(C0).cuda()
                dBr = torch.from_numpy(np.ascontiguousarray(B.reshape(N, K).T)).cuda()
                ok = []
                with api.Engine(0) as e:
                    e.set_matrix_csr(r1 - r0, K, lrp, lci, lv)
                    for nchunks in (1, 3):
                        e.dist_prepare(comm, world, rank, ranges, N, nchunks=nchunks, form=0, stream=st)
                        out = torch.full((M * N,), float("nan"), device="cuda")
                        e.dist_spmm(comm, world, rank, ranges, N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr(), M, out.data_ptr(), M, nchunks=nchunks, stream=st)
                        torch.cuda.current_stream().synchronize()
                        ok.append(bool(np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))))
                    cin = torch.from_numpy(np.ascontiguousarray(C0.reshape(N, M).T)).cuda()
                    outr = torch.full((M, N), float("nan"), device="cuda")
                    e.dist_spmm_rm(comm, world, rank, ranges, N, ALPHA, dBr.data_ptr(), N, BETA, cin.data_ptr(), N, outr.data_ptr(), N, stream=st)
                    torch.cuda.current_stream().synchronize()
                    ok.append(bool(np.array_equal(outr.cpu().numpy().view(np.uint32), want_rm.view(np.uint32))))
                return ok

            res = Ranks(world).run(rank_fn)
            assert all(all(r) for r in res), (N, mode, ranges, res)
            out_all[(N, mode)] = ranges
    return f"ranks without rows complete every form: {out_all[(16, 'nnz')]}"


SCENARIOS = {
    "empty_ranges": scenario_empty_ranges,
    "hub_rows": scenario_hub_rows,
    "config5_full": scenario_config5_full,
    "config4_full": scenario_config4_full,
    "colmajor": scenario_colmajor,
    "colmajor_clustered": lambda w: scenario_colmajor(w, clustered=True),
    "rowmajor": scenario_rowmajor,
    "bell": scenario_bell,
    "errors": scenario_errors,
}


def main(argv):
    scenario, world = argv[1], int(argv[2])
    from sextans_amd import api
    if api.device_count() < 1:
        raise SystemExit("loopback worker: no gfx950 device (the engine has no CPU path)")
    api.dist_bind_library(fake_rccl_path())
    msg = SCENARIOS[scenario](world)
    print(f"LOOPBACK OK {scenario} world={world}: {msg}", flush=True)


if __name__ == "__main__":
    main(sys.argv)
