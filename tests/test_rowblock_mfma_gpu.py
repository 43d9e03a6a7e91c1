"""fp32 matrix-core path for dense row blocks ("mfma_dense_tiles" = 2) and the documented accuracy modes (tasks 3 and 5).

north_star: "feeds MFMA only where a tile is actually dense".  The reference's PEs multiply and accumulate in fp32
(/root/reference/src/sextans.cpp:285-295, 425-446); v_mfma_f32_16x16x4_f32 is a k-ordered chain of fused multiply-adds, so a row routed
to it and walked in ascending column order gets EXACTLY the bits of the engine's "exact" = 0 kernels.  The tests demand
  (a) bit equality with the CPU statement of that arithmetic (oracle.spmm_fma: cpu_spmm_CSR's loop nest with fmaf),
  (b) bit equality with the engine's own "exact" = 0 path without routing,
  (c) the bound of SEXTANS_MODE_FAST against the PINNED oracle (cpu_spmm_CSR): |d| <= 1e-4 * (|alpha| sum|a b| + |beta c|)."""
import numpy as np
import pytest

from util import ALPHA, BETA, random_csr

pytestmark = pytest.mark.gpu


def _dense_blocks(rs, nbr, nbc, blocks_per_row, bs=32):
    """Block-banded-ish matrix of fully dense bs x bs blocks (CSR, ascending columns)."""
    M, K = nbr * bs, nbc * bs
    rows = []
    for br in range(nbr):
        cols = np.sort(rs.choice(nbc, size=min(blocks_per_row, nbc), replace=False))
        c = (cols[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)
        for _ in range(bs):
            rows.append(c)
    lens = np.array([len(r) for r in rows])
    rp = np.zeros(M + 1, np.int32); rp[1:] = np.cumsum(lens)
    ci = np.concatenate(rows).astype(np.int32)
    v = rs.uniform(-1, 1, len(ci)).astype(np.float32)
    return rp, ci, v, M, K


def _bound(o, M, N, K, rp, ci, v, B, C0):
    """1e-4 * (|alpha| sum|a b| + |beta c|) per element, through the oracle on absolute values."""
    s = np.zeros(M * N, np.float32)
    o.spmm(M, N, K, np.float32(1.0), rp, ci, np.abs(v), np.abs(B), np.float32(0.0), s)
    return 1e-4 * (abs(float(ALPHA)) * s.astype(np.float64) + np.abs(float(BETA) * C0.astype(np.float64))) + 1e-30


def _run(e, M, N, K, B, C0, rp_time=1):
    out = C0.copy()
    e.spmm(N, ALPHA, B, BETA, out, rp_time=rp_time)
    return out


@pytest.mark.parametrize("N", [8, 16, 24, 32, 64, 128, 136])
def test_dense_blocks_bitwise_equal_to_the_fma_chain(sx, oracle, N):
    rs = np.random.RandomState(N)
    rp, ci, v, M, K = _dense_blocks(rs, 40, 50, 6)
    M_full = M
    # a ragged tail: 9 more rows (not a full block of 16: they stay on the CSR kernels)
    extra_rp, extra_ci, extra_v = random_csr(rs, 9, K, 20)
    rp = np.concatenate([rp, rp[-1] + extra_rp[1:]]).astype(np.int32)
    ci = np.concatenate([ci, extra_ci]).astype(np.int32); v = np.concatenate([v, extra_v]).astype(np.float32)
    M += 9
    B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm_fma(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
    pinned = C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, pinned)
    with sx.Engine(0) as e:
        e.set_option("exact", 0)
        e.set_matrix_csr(M, K, rp, ci, v)
        plain = _run(e, M, N, K, B, C0)                      # "exact" = 0 on the CSR kernels alone
        assert np.array_equal(plain.view(np.uint32), want.view(np.uint32)), e.last_kernel()
        e.set_option("mfma_dense_tiles", 2)
        got = _run(e, M, N, K, B, C0)
        assert "rowblock_mfma_f32" in e.last_kernel(), e.last_kernel()
        assert int(e.get_stat("dense_tiles")) == M_full // 16 and abs(e.get_stat("dense_tile_fraction") - (rp[M_full] / rp[-1])) < 1e-9
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (N, e.last_kernel())
        assert np.array_equal(_run(e, M, N, K, B, C0, rp_time=3).view(np.uint32), want.view(np.uint32))   # hipGraph replay
        assert np.all(np.abs(got.astype(np.float64) - pinned) <= _bound(oracle, M, N, K, rp, ci, v, B, C0))
        e.set_option("mfma_dense_tiles", 0)                  # and back: the rows return to the CSR kernels
        assert np.array_equal(_run(e, M, N, K, B, C0).view(np.uint32), want.view(np.uint32))
        assert "rowblock" not in e.last_kernel()


@pytest.mark.parametrize("dof,thr", [(6, 50), (3, 30), (3, 50)])
def test_fem_row_blocks_partial_routing(sx, oracle, dof, thr):
    """A 3-D FEM matrix: 6-dof node blocks fill their 16 x 4 fragments to ~0.6 (routed at the default threshold), 3-dof ones to ~0.35
    (routed at 30 %, not at 50 %); boundary blocks differ from interior ones, so routed and unrouted rows share the call."""
    from sextans_amd import api
    nx, ny, nz = 13, 11, 9
    rp, ci, v = api.gen_fem3d_host(nx, ny, nz, dof, 3)
    M = K = nx * ny * nz * dof
    rs = np.random.RandomState(dof)
    for N in (16, 128):
        B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
        want = C0.copy()
        oracle.spmm_fma(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
        with sx.Engine(0) as e:
            e.set_option("exact", 0)
            e.set_option("dense_tile_fill_x100", thr)
            e.set_option("mfma_dense_tiles", 2)
            e.set_matrix_csr(M, K, rp, ci, v)
            got = _run(e, M, N, K, B, C0)
            frac = e.get_stat("dense_tile_fraction")
            if (dof, thr) == (3, 50):
                assert frac < 0.05 and "rowblock" not in e.last_kernel()
            else:
                assert frac > 0.5 and "rowblock_mfma_f32" in e.last_kernel(), (frac, e.last_kernel())
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (dof, thr, N, e.last_kernel(), frac)


def test_unsorted_rows_and_duplicates_are_not_routed(sx, oracle):
    """The chain order of a row IS its CSR order: a block with a row whose columns are not strictly ascending (unsorted, or a duplicate
    (row, column) pair -- the loader keeps those, sparse_helper.h:112-167) stays on the CSR kernels; its neighbours are routed."""
    rs = np.random.RandomState(9)
    rp, ci, v, M, K = _dense_blocks(rs, 6, 8, 3)
    ci = ci.copy()
    a, b = int(rp[5]), int(rp[6])
    ci[a:b] = ci[a:b][::-1]                                   # row 5 (block 0): descending columns
    a = int(rp[40])
    ci[a + 1] = ci[a]                                         # row 40 (block 2): a duplicate pair
    N = 32
    B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm_fma(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
    with sx.Engine(0) as e:
        e.set_option("exact", 0)
        e.set_option("mfma_dense_tiles", 2)
        e.set_matrix_csr(M, K, rp, ci, v)
        got = _run(e, M, N, K, B, C0)
        assert int(e.get_stat("dense_tiles")) == M // 16 - 2
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_row_ranges_and_rowmajor_calls(sx, oracle):
    """Row-range calls (the chunks of the multi-GPU pipeline) cut through routed blocks; the row-major entry point reaches the path
    through its column-major copies.  Same bits."""
    import torch
    rs = np.random.RandomState(4)
    rp, ci, v, M, K = _dense_blocks(rs, 12, 12, 4)
    N = 48
    B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm_fma(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
    st = torch.cuda.current_stream().cuda_stream
    with sx.Engine(0) as e:
        e.set_option("exact", 0)
        e.set_option("mfma_dense_tiles", 2)
        e.set_matrix_csr(M, K, rp, ci, v)
        dB = torch.from_numpy(B).cuda(); dC = torch.from_numpy(C0).cuda()
        out = torch.full((M * N,), float("nan"), device="cuda")
        cuts = [0, 37, 200, 201, M]                           # not multiples of 16
        for i in range(len(cuts) - 1):
            c0, c1 = cuts[i], cuts[i + 1]
            e.spmm_device_rows(N, ALPHA, dB.data_ptr(), K, BETA, dC.data_ptr() + 4 * c0, M, out.data_ptr() + 4 * c0, M, c0, c1,
                               reuse_b_panels=i > 0, stream=st)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
        # sextans_dist_spmm without a communicator: 3 chunks of this rank's "slab"
        out = torch.full((M * N,), float("nan"), device="cuda")
        e.dist_spmm(None, 1, 0, [(0, M)], N, ALPHA, dB.data_ptr(), K, BETA, dC.data_ptr(), M, out.data_ptr(), M, nchunks=3, stream=st)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
        # row-major operands
        Br = torch.from_numpy(np.ascontiguousarray(B.reshape(N, K).T)).cuda(); Cr = torch.from_numpy(np.ascontiguousarray(C0.reshape(N, M).T)).cuda()
        outr = torch.empty_like(Cr)
        e.spmm_device_rm(N, ALPHA, Br.data_ptr(), N, BETA, Cr.data_ptr(), N, outr.data_ptr(), N, st)
        torch.cuda.synchronize()
        assert np.array_equal(np.ascontiguousarray(outr.cpu().numpy().T).reshape(-1).view(np.uint32), want.view(np.uint32))


def test_mode_switch_and_its_guarantee(sx, oracle):
    """SEXTANS_MODE_FAST = "exact" 0 + "split_rows" -1 (+ here, by hand, "mfma_dense_tiles" 2 on every second case: bit-identical to the
    mode, so it must meet the same bound); SEXTANS_MODE_STRICT restores bit identity with cpu_spmm_CSR.  One fuzz over the plan forms a matrix can take -- dense blocks, FEM bricks, a mesh in a random node order (graph
    clustering), uniformly random rows (gather kernel), power-law rows (hub pieces), mixed plans -- in FAST mode against the bound,
    column-major and row-major entry points."""
    import torch
    from sextans_amd import api, meshgen
    rs = np.random.RandomState(12)
    st = torch.cuda.current_stream().cuda_stream
    cases = []
    rp, ci, v, M, K = _dense_blocks(rs, 20, 24, 5); cases.append(("dense blocks", rp, ci, v, M, K))
    rp, ci, v = api.gen_fem3d_host(16, 14, 12, 3, 5); M = K = 16 * 14 * 12 * 3; cases.append(("fem", rp, ci, v, M, K))
    rp2, ci2, v2 = meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M // 3, 3, 9)); cases.append(("fem random order", rp2, ci2, v2, M, K))
    rp, ci, v = random_csr(rs, 3000, 2500, 11, long_rows=2); cases.append(("random", rp, ci, v, 3000, 2500))
    rp, ci, v = api.gen_powerlaw_host(40_000, 40_000, 4, 120, 30_000, 3); cases.append(("power law", rp, ci, v, 40_000, 40_000))
    rp, ci, v = api.gen_fem3d_host(12, 10, 8, 6, 2); M = K = 12 * 10 * 8 * 6; cases.append(("fem 6 dof", rp, ci, v, M, K))
    rp, ci, v = api.gen_stencil2d_host(120, 110, 5, 1, 4); cases.append(("5-point stencil (lane-per-row kernel)", rp, ci, v, 120 * 110, 120 * 110))
    frp, fci, fv = api.gen_fem3d_host(14, 12, 10, 3, 8); Mf = 14 * 12 * 10 * 3                      # mixed plan: mesh rows + rows without reuse
    rrp, rci, rv = random_csr(rs, 2000, Mf, 30)
    cases.append(("mixed plan", np.concatenate([frp, frp[-1] + rrp[1:]]).astype(np.int32), np.concatenate([fci, rci]).astype(np.int32),
                  np.concatenate([fv, rv]).astype(np.float32), Mf + 2000, Mf))
    with sx.Engine(0) as e:
        assert e.get_option("mode") == 0
        e.set_option("mode", 1)
        assert (e.get_option("exact"), e.get_option("split_rows"), e.get_option("mfma_dense_tiles"), e.get_option("mode")) == (0, -1, 0, 1)
        e.set_option("exact", 1)
        assert e.get_option("mode") == -1                    # set apart by hand
        e.set_option("mode", 0)
        assert (e.get_option("exact"), e.get_option("split_rows"), e.get_option("mfma_dense_tiles"), e.get_option("mode")) == (1, 0, 0, 0)
        with pytest.raises(api.SextansError):
            e.set_option("mode", 2)
        worst = 0.0
        for name, rp, ci, v, M, K in cases:
            for N in (16, 40, 128):
                B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
                pinned = C0.copy()
                oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, pinned)
                bound = _bound(oracle, M, N, K, rp, ci, v, B, C0)
                e.set_option("mode", 0)
                e.set_matrix_csr(M, K, rp, ci, v)
                strict = _run(e, M, N, K, B, C0)
                assert np.array_equal(strict.view(np.uint32), pinned.view(np.uint32)), (name, N, e.last_kernel())
                e.set_option("mode", 1)
                fast = _run(e, M, N, K, B, C0)
                k_fast = e.last_kernel()
                if name in ("dense blocks", "fem 6 dof"):   # dense row blocks on the fp32 matrix cores: not one bit moves
                    e.set_option("mfma_dense_tiles", 2)
                    routed = _run(e, M, N, K, B, C0)
                    assert "rowblock_mfma_f32" in e.last_kernel() and np.array_equal(routed.view(np.uint32), fast.view(np.uint32)), (name, N)
                ratio = float(np.max(np.abs(fast.astype(np.float64) - pinned) / bound))
                worst = max(worst, ratio)
                assert ratio <= 1.0, (name, N, k_fast, ratio)
                assert np.array_equal(_run(e, M, N, K, B, C0).view(np.uint32), fast.view(np.uint32))       # deterministic
                # row-major entry point, same mode
                Br = torch.from_numpy(np.ascontiguousarray(B.reshape(N, K).T)).cuda(); Cr = torch.from_numpy(np.ascontiguousarray(C0.reshape(N, M).T)).cuda()
                outr = torch.empty_like(Cr)
                e.spmm_device_rm(N, ALPHA, Br.data_ptr(), N, BETA, Cr.data_ptr(), N, outr.data_ptr(), N, st)
                torch.cuda.synchronize()
                fr = np.ascontiguousarray(outr.cpu().numpy().T).reshape(-1)
                assert float(np.max(np.abs(fr.astype(np.float64) - pinned) / bound)) <= 1.0, (name, N, e.last_kernel())
        print(f"SEXTANS_MODE_FAST: worst |fast - cpu_spmm_CSR| / (1e-4 * (|alpha| sum|a b| + |beta c|)) = {worst:.2e}")
        assert worst < 0.05      # fp32 roundoff of another fp32 summation: orders of magnitude inside the guarantee


@pytest.mark.parametrize("K", [37, 38, 39, 64])
def test_last_column_group_of_a_K_that_is_no_multiple_of_4(sx, oracle, K):
    """A group covers 4 consecutive columns; the last group of K = 4 k + r reaches past row K - 1 of a B panel -- into the next panel or into
    workspace memory no repack ever wrote.  The kernel's buffer resources end with the panel, so those rows read as 0 (0 x NaN would not be 0):
    the workspace is first filled with NaNs by a wider call on a NaN B, then the real call must still give the fmaf chain bit for bit."""
    rs = np.random.RandomState(K)
    M = 80
    dense = rs.rand(M, K) < 0.8
    dense[:, K - 1] = True                                    # the last column is used by every row
    rp = np.zeros(M + 1, np.int32); rp[1:] = np.cumsum(dense.sum(1))
    ci = np.concatenate([np.nonzero(dense[r])[0] for r in range(M)]).astype(np.int32)
    v = rs.uniform(-1, 1, len(ci)).astype(np.float32)
    N = 24
    B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm_fma(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
    with sx.Engine(0) as e:
        e.set_option("exact", 0)
        e.set_option("mfma_dense_tiles", 2)
        e.set_matrix_csr(M, K, rp, ci, v)
        poison = np.full(K * 64, np.nan, np.float32)
        e.spmm(64, ALPHA, poison, BETA, np.zeros(M * 64, np.float32))        # every panel of the workspace now holds NaNs
        got = _run(e, M, N, K, B, C0)
        assert "rowblock_mfma_f32" in e.last_kernel() and int(e.get_stat("dense_tiles")) == M // 16
        assert not np.isnan(got).any()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), K
