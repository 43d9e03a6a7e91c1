"""(Round 5: + unsymmetric patterns and rectangular matrices -- clustered over A + A^T / the row-similarity graph -- and every trial once
more through the ROW-major entry point sextans_spmm_device_rm with padded leading dimensions.)
Randomised parity of the round-4 plan forms: grid / mesh matrices of odd sizes, with rows removed or emptied, in natural, random and
partially shuffled numberings, at N that are and are not multiples of 16, under random settings of the options that select a plan form
(row_cluster, share_index, row_sets, small_panel, relabel_columns, refine_sweeps, kernel, lanes_per_row).  Every result is compared
bit for bit with cpu_spmm_CSR (oracle/), whole-matrix calls and rp_time replays alike.  The point is the corners no hand-written case
names: partial bricks, blocks cut by the dictionary capacity, chains of shared index lists that end at a block boundary, empty slots
inside a row set, rows longer than the register-resident batches next to rows of one entry."""
import os

import numpy as np
import pytest

from util import ALPHA, BETA

pytestmark = pytest.mark.gpu

DEFAULTS = dict(row_cluster=-1, share_index=1, row_sets=2, small_panel=1, relabel_columns=1, refine_sweeps=8, kernel=0, lanes_per_row=0,
                panel_min_reuse_x100=200, panel_min_reuse_wide_x100=150, fuse_b=0)


def _matrix(rs):
    from sextans_amd import api, meshgen
    kind = rs.randint(0, 5)
    if kind == 0:
        dims = [int(rs.randint(3, 22)) for _ in range(3)]
        dof = int(rs.choice([1, 1, 2, 3, 4]))
        rp, ci, v = api.gen_fem3d_host(*dims, dof, int(rs.randint(1, 99)))
        M, name = dims[0] * dims[1] * dims[2] * dof, f"fem {dims} x {dof}"
    elif kind == 1:
        nx, ny = int(rs.randint(5, 150)), int(rs.randint(5, 120))
        pts, dof = int(rs.choice([5, 9])), int(rs.choice([1, 2, 3]))
        rp, ci, v = api.gen_stencil2d_host(nx, ny, pts, dof, int(rs.randint(1, 99)))
        M, name = nx * ny * dof, f"stencil {nx}x{ny} {pts}-pt x {dof}"
    elif kind == 2:
        n = int(rs.randint(4, 16))
        dof = int(rs.choice([1, 3]))
        rp, ci, v, M = meshgen.jittered_mesh3d(n, n + 1, n + 2, int(rs.randint(1, 99)), numbering=str(rs.choice(["sweep", "random", "grid"])), dof=dof)
        name = f"mesh {n} x {dof}"
    elif kind == 3:   # a grid matrix under a random node renumbering
        dims = [int(rs.randint(4, 18)) for _ in range(3)]
        dof = int(rs.choice([1, 3]))
        rp, ci, v = api.gen_fem3d_host(*dims, dof, 5)
        M = dims[0] * dims[1] * dims[2] * dof
        rp, ci, v = meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M // dof, dof, int(rs.randint(1, 99))))
        name = f"fem {dims} x {dof} random order"
    else:             # a grid matrix with a shuffled slab in the middle
        dims = [int(rs.randint(6, 20)) for _ in range(3)]
        rp, ci, v = api.gen_fem3d_host(*dims, 1, 5)
        M = dims[0] * dims[1] * dims[2]
        perm = np.arange(M)
        a, b = sorted(rs.randint(0, M, 2))
        perm[a:b] = perm[a:b][rs.permutation(b - a)]
        rp, ci, v = meshgen.permute_symmetric(rp, ci, v, M, perm)
        name = f"fem {dims} shuffled [{a},{b})"
    rp, ci, v = np.array(rp, np.int32), np.array(ci, np.int32), np.array(v, np.float32)
    lens = np.diff(rp)
    keep = np.ones(len(ci), bool)
    K = M
    mode = rs.randint(0, 6)
    if mode == 1:     # a few rows emptied
        for r in rs.randint(0, M, max(1, M // 50)):
            keep[rp[r]:rp[r + 1]] = False
    elif mode == 2:   # random entries dropped (rows of a node stop sharing their index lists)
        keep &= rs.rand(len(ci)) > 0.1
    elif mode == 3:   # one row made long (dense-ish) -- columns stay sorted and distinct
        r = int(rs.randint(0, M))
        extra = np.setdiff1d(rs.choice(M, min(M, 200), replace=False), ci[rp[r]:rp[r + 1]]).astype(np.int32)
        row_c = np.concatenate([ci[rp[r]:rp[r + 1]], extra]); order = np.argsort(row_c, kind="stable")
        row_v = np.concatenate([v[rp[r]:rp[r + 1]], rs.uniform(-1, 1, len(extra)).astype(np.float32)])[order]
        ci = np.concatenate([ci[:rp[r]], row_c[order], ci[rp[r + 1]:]]); v = np.concatenate([v[:rp[r]], row_v, v[rp[r + 1]:]])
        lens[r] += len(extra)
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        keep = np.ones(len(ci), bool)
    elif mode == 4:   # unsymmetric pattern: 30 % of the strictly lower entries dropped (clustered over A + A^T)
        rows = np.repeat(np.arange(M), lens)
        keep &= ~((ci < rows) & (rs.rand(len(ci)) < 0.3))
    elif mode == 5:   # rectangular: a random third of the columns dropped, the rest renumbered (clustered over the row-similarity graph)
        colkeep = rs.rand(M) > 0.33
        colkeep[int(rs.randint(0, M))] = True
        newcol = np.cumsum(colkeep) - 1
        keep &= colkeep[ci]
        ci = newcol[ci].astype(np.int32)
        K = int(colkeep.sum())
    if not keep.all():
        rows = np.repeat(np.arange(M), lens)
        lens = np.bincount(rows[keep], minlength=M)
        ci, v = ci[keep], v[keep]
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    return name + f" mode {mode}", M, K, rp, ci.astype(np.int32), v.astype(np.float32)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SEXTANS_FUZZ_SEEDS", "256"))))   # (a soak run: SEXTANS_FUZZ_SEEDS=3000)
def test_random_plan_forms_are_bit_identical(engine, oracle, seed):
    rs = np.random.RandomState(1000 + seed)
    import torch
    name, M, K, rp, ci, v = _matrix(rs)
    try:
        for trial in range(4):
            N = int(rs.choice([8, 16, 16, 24, 32, 40, 64, 136]))
            opts = dict(DEFAULTS, row_cluster=int(rs.choice([-1, 0, 1, 1, 2, 2])), share_index=int(rs.choice([0, 1, 1])), row_sets=int(rs.choice([1, 2, 3])),
                        small_panel=int(rs.choice([0, 1])), relabel_columns=int(rs.choice([0, 1, 1])), refine_sweeps=int(rs.choice([0, 8])),
                        kernel=int(rs.choice([0, 0, 0, 2, 4])), lanes_per_row=int(rs.choice([0, 0, 4])))
            for k, val in opts.items():
                engine.set_option(k, val)
            engine.set_matrix_csr(M, K, rp, ci, v)
            B = rs.uniform(-1, 1, K * N).astype(np.float32)
            C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
            want = C0.copy()
            oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
            for rp_time in (1, 3):
                out = C0.copy()
                engine.spmm(N, ALPHA, B, BETA, out, rp_time=rp_time)
                assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (name, N, opts, rp_time, engine.last_kernel())
            # the same through the row-major entry point (padded leading dimensions, separate C_out)
            ldb, ldi, ldo = N + 4 * int(rs.randint(0, 3)), N + 4 * int(rs.randint(0, 3)), N + 4 * int(rs.randint(0, 3))
            tb = torch.zeros((K, ldb), device="cuda"); tb[:, :N] = torch.from_numpy(np.ascontiguousarray(B.reshape(N, K).T)).cuda()
            ti = torch.zeros((M, ldi), device="cuda"); ti[:, :N] = torch.from_numpy(np.ascontiguousarray(C0.reshape(N, M).T)).cuda()
            to = torch.full((M, ldo), -3.0, device="cuda")
            engine.spmm_device_rm(N, float(ALPHA), tb.data_ptr(), ldb, float(BETA), ti.data_ptr(), ldi, to.data_ptr(), ldo, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            got = np.ascontiguousarray(to[:, :N].cpu().numpy().T).reshape(-1)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (name, N, opts, "row-major", engine.last_kernel())
            assert ldo == N or bool((to[:, N:] == -3.0).all()), (name, N, "columns beyond N written")
            if os.environ.get("SEXTANS_FUZZ_VERBOSE"):
                print(f"[fuzz] {name} N={N} rc={opts['row_cluster']}: {engine.last_kernel()} state={int(engine.get_stat('row_cluster'))} "
                      f"sets={int(engine.get_stat('row_sets'))} idx/val={engine.get_stat('index_stream_entries') / max(engine.get_stat('value_stream_entries'), 1):.2f}")
            if trial == 3:
                # Round 6: the SAME plan form with fused multiply-adds ("exact" = 0, the arithmetic of SEXTANS_MODE_FAST; rows stay whole, so
                # nothing is re-associated) -- bit-identical to the oracle's fmaf chain whatever kernel the options selected, with and without
                # dense row blocks routed to the fp32 matrix cores, through both entry points.
                want_fma = C0.copy()
                oracle.spmm_fma(M, N, K, ALPHA, rp, ci, v, B, BETA, want_fma)
                engine.set_option("exact", 0)
                for tiles in (0, 2):
                    engine.set_option("dense_tile_fill_x100", int(rs.choice([20, 35, 50])))
                    engine.set_option("mfma_dense_tiles", tiles)
                    out = C0.copy()
                    engine.spmm(N, ALPHA, B, BETA, out)
                    assert np.array_equal(out.view(np.uint32), want_fma.view(np.uint32)), (name, N, opts, "exact=0", tiles, engine.last_kernel())
                    to.fill_(-3.0)
                    engine.spmm_device_rm(N, float(ALPHA), tb.data_ptr(), ldb, float(BETA), ti.data_ptr(), ldi, to.data_ptr(), ldo, torch.cuda.current_stream().cuda_stream)
                    torch.cuda.synchronize()
                    got = np.ascontiguousarray(to[:, :N].cpu().numpy().T).reshape(-1)
                    assert np.array_equal(got.view(np.uint32), want_fma.view(np.uint32)), (name, N, opts, "exact=0 row-major", tiles, engine.last_kernel())
    finally:
        for k, val in dict(DEFAULTS, exact=1, mfma_dense_tiles=0, dense_tile_fill_x100=50).items():
            engine.set_option(k, val)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SEXTANS_FUZZ_LARGE_SEEDS", "8"))))
def test_random_large_matrices_default_options(engine, oracle, seed):
    """The same at sizes where B no longer fits the L2s (10^5 .. 10^6 rows): the automatic choices of the dispatcher -- grid bricks, graph
    clustering + reordered form, natural plan, lane-per-row kernel -- under DEFAULT options, whole-matrix calls and one row-range call."""
    from sextans_amd import api, meshgen
    rs = np.random.RandomState(7000 + seed)
    kind = seed % 4
    if kind == 0:
        dims, dof = [int(rs.randint(35, 60)) for _ in range(3)], int(rs.choice([1, 3]))
        rp, ci, v = api.gen_fem3d_host(*dims, dof, 3)
        M = dims[0] * dims[1] * dims[2] * dof
    elif kind == 1:
        dims, dof = [int(rs.randint(30, 50)) for _ in range(3)], int(rs.choice([1, 3]))
        rp, ci, v = api.gen_fem3d_host(*dims, dof, 3)
        M = dims[0] * dims[1] * dims[2] * dof
        rp, ci, v = meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M // dof, dof, seed))
    elif kind == 2:
        n = int(rs.randint(40, 60))
        rp, ci, v, M = meshgen.jittered_mesh3d(n, n, n, seed, numbering=str(rs.choice(["sweep", "random"])), dof=int(rs.choice([1, 3])))
    else:
        nx, ny, pts = int(rs.randint(300, 700)), int(rs.randint(300, 600)), int(rs.choice([5, 9]))
        rp, ci, v = api.gen_stencil2d_host(nx, ny, pts, 1, 3)
        M = nx * ny
    rp, ci, v = np.array(rp, np.int32), np.array(ci, np.int32), np.array(v, np.float32)
    K = M
    for k, val in DEFAULTS.items():
        engine.set_option(k, val)
    engine.set_matrix_csr(M, K, rp, ci, v)
    for N in (16, int(rs.choice([8, 24, 32, 48]))):
        B = rs.uniform(-1, 1, K * N).astype(np.float32)
        C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
        want = C0.copy()
        oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
        out = C0.copy()
        engine.spmm(N, ALPHA, B, BETA, out)
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (kind, M, N, engine.last_kernel(), engine.get_stat("row_cluster"))
        # one row-range call (what a rank of the row-partitioned SpMM issues): its rows into a packed slab, the natural-order plan
        import torch
        st = torch.cuda.current_stream().cuda_stream
        c0, c1 = sorted(int(x) for x in rs.randint(0, M, 2))
        if c1 > c0:
            dB = torch.from_numpy(B).cuda(); dCin = torch.from_numpy(C0).cuda()
            slab = torch.full(((c1 - c0) * N,), float("nan"), device="cuda")
            engine.spmm_device_rows(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr() + 4 * c0, M, slab.data_ptr(), c1 - c0, c0, c1, stream=st)
            torch.cuda.synchronize()
            got = slab.cpu().numpy().reshape(N, c1 - c0)
            assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(want.reshape(N, M)[:, c0:c1]).view(np.uint32)), (kind, M, N, c0, c1, engine.last_kernel())
            del dB, dCin, slab
