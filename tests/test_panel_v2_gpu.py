"""spmm_csr_panel_v2 (csrc/spmm_panel_v2.h): the LDS-panel kernel with the row entries register-resident and the tile loop
inside the workgroup, in two shapes: H = 1 (16-column tiles, 4 workgroups per CU, panels by LDS-DMA; the default for
every N) and H = 2 (8 columns per lane, 32-column super tiles, 2 workgroups per CU; option cols_per_lane = 8).  Same packed plan, same per-row order as spmm_csr_panel:
every result must be bit-identical to cpu_spmm_CSR (oracle), for every way the dispatcher can cut N into super tiles,
odd 16-column tiles and 8-column remainders, with LDS-DMA staging from repacked panels and with register staging from
column-major B, on whole matrices and on block-aligned row ranges, with and without long rows on the piece path."""
import numpy as np
import pytest

from util import ALPHA, BETA, random_csr

pytestmark = pytest.mark.gpu

BASE = dict(lanes_per_row=4, stage_a=1, xcd_remap=1, exact=1, kernel=0, panel_min_reuse_x100=200, fuse_b=0,
            split_rows=0, bucket_rows=0, cols_per_lane=0, tiles_per_wg=0, panel_v2=-1)


def _set(engine, **opts):
    d = dict(BASE)
    d.update(opts)
    for k, v in d.items():
        engine.set_option(k, v)


@pytest.fixture(scope="module")
def fem():
    from sextans_amd import api
    rp, ci, v = api.gen_fem3d_host(12, 11, 10, 3, 7)
    return 12 * 11 * 10 * 3, rp, ci, v


@pytest.mark.parametrize("N", [32, 48, 64, 72, 128, 136, 256])
@pytest.mark.parametrize("fuse_b", [0, 1])
def test_wide_matches_oracle_for_every_cut_of_n(engine, oracle, fem, N, fuse_b):
    M, rp, ci, v = fem
    K = M
    rs = np.random.RandomState(N + fuse_b)
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
    _set(engine, fuse_b=fuse_b)
    engine.set_matrix_csr(M, K, rp, ci, v)
    try:
        for tpw, cpl in ((0, 8), (1, 8), (2, 8), (3, 8), (0, 0), (3, 4)):
            for exact in (1, 0):
                _set(engine, fuse_b=fuse_b, tiles_per_wg=tpw, exact=exact, cols_per_lane=cpl, panel_v2=1 - fuse_b if cpl != 8 else -1)
                out = C0.copy()
                engine.spmm(N, ALPHA, B, BETA, out)
                assert engine.last_kernel() == ("spmm_csr_panel" if fuse_b and cpl != 8 else "spmm_csr_panel_v2"), (tpw, cpl, engine.last_kernel())
                if exact:
                    assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (N, fuse_b, tpw)
                else:       # FMA variant: the stated 1e-4 relative bound
                    assert np.allclose(out, want, rtol=1e-4, atol=1e-4), (N, fuse_b, tpw)
        # panel_v2 = 0 switches the new form off: the round-1 panel kernel, same bits
        _set(engine, fuse_b=fuse_b, panel_v2=0)
        out = C0.copy()
        engine.spmm(N, ALPHA, B, BETA, out)
        assert engine.last_kernel() == "spmm_csr_panel"
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    finally:
        _set(engine)


def test_wide_rp_time_loop_and_alpha_beta_edge_values(engine, oracle, fem):
    M, rp, ci, v = fem
    K, N = M, 64
    rs = np.random.RandomState(9)
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    _set(engine)
    engine.set_matrix_csr(M, K, rp, ci, v)
    for alpha, beta in ((1.0, 0.0), (0.0, 1.0), (-0.5, 2.0), (float(ALPHA), float(BETA))):
        want = C0.copy()
        oracle.spmm(M, N, K, np.float32(alpha), rp, ci, v, B, np.float32(beta), want)
        out = C0.copy()
        engine.spmm(N, alpha, B, beta, out, rp_time=5)      # hipGraph replay; panels repacked once
        assert engine.last_kernel() == "spmm_csr_panel_v2"
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (alpha, beta)


def test_wide_on_block_aligned_row_ranges_and_with_long_rows(engine, oracle, fem):
    """Row-range calls (multi-GPU chunks) keep the wide kernel when cut at row-block boundaries; long rows leave through
    the piece path (skip flags) and are folded back in order."""
    import torch
    M, rp, ci, v = fem
    K, N = M, 96
    rs = np.random.RandomState(21)
    hub = 2000
    cols = np.sort(rs.choice(K, 700, replace=False)).astype(np.int32)
    a, b = int(rp[hub]), int(rp[hub + 1])
    ci2 = np.concatenate([ci[:a], cols, ci[b:]]).astype(np.int32)
    v2 = np.concatenate([v[:a], rs.uniform(-1, 1, 700).astype(np.float32), v[b:]]).astype(np.float32)
    rp2 = rp.copy(); rp2[hub + 1:] += 700 - (b - a)
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp2, ci2, v2, B, BETA, want)
    _set(engine, bucket_rows=300)
    try:
        engine.set_matrix_csr(M, K, rp2, ci2, v2)
        assert engine.get_stat("piece_path_rows") == 1
        out = C0.copy()
        engine.spmm(N, ALPHA, B, BETA, out)
        assert engine.last_kernel() == "spmm_csr_panel_v2+hub_pieces"
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
        st = torch.cuda.current_stream().cuda_stream
        dB = torch.from_numpy(B).cuda(); dCin = torch.from_numpy(C0).cuda()
        cuts = [0, engine.align_row(N, M // 3), engine.align_row(N, 2 * M // 3), M]
        got = torch.full((M * N,), float("nan"), device="cuda")
        for i in range(3):
            c0, c1 = cuts[i], cuts[i + 1]
            slab = torch.full(((c1 - c0) * N,), float("nan"), device="cuda")
            engine.spmm_device_rows(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr() + 4 * c0, M, slab.data_ptr(), c1 - c0,
                                    c0, c1, reuse_b_panels=i > 0, stream=st)
            assert engine.last_kernel().startswith("spmm_csr_panel_v2"), engine.last_kernel()
            got.view(N, M)[:, c0:c1] = slab.view(N, c1 - c0)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
    finally:
        _set(engine)


@pytest.mark.parametrize("seed", range(8))
def test_wide_on_random_structures(engine, oracle, seed):
    """Banded random matrices (dictionary-only plans of every dictionary size up to the cap), ragged row lengths incl.
    empty rows and rows longer than a batch ring, M not a multiple of anything."""
    rs = np.random.RandomState(100 + seed)
    M = int(rs.randint(300, 3000))
    K = int(rs.randint(200, 5000))
    bw = int(rs.choice([40, 150, 400]))
    rows = []
    for r in range(M):
        n = int(rs.choice([0, 1, 3, 4, 5, 16, 17, 47, 48, 49, 64, 65, 95, 96, 97, 100, 130])) if rs.rand() < 0.5 else int(rs.randint(0, 30))
        lo = max(0, min(K - 1, int(r * K / M) - bw)); hi = min(K, lo + 2 * bw + 1)
        n = min(n, hi - lo)
        c = np.sort(rs.choice(np.arange(lo, hi), n, replace=False)).astype(np.int32)
        rows.append(c)
    rp = np.zeros(M + 1, np.int32); rp[1:] = np.cumsum([len(c) for c in rows])
    ci = np.concatenate(rows).astype(np.int32) if rp[-1] else np.zeros(0, np.int32)
    v = rs.uniform(-1, 1, len(ci)).astype(np.float32)
    N = int(rs.choice([32, 40, 64, 96]))
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
    fuse_b = int(rs.randint(0, 2))
    _set(engine, kernel=2, panel_min_reuse_x100=0, fuse_b=fuse_b, tiles_per_wg=int(rs.randint(0, 3)),
         cols_per_lane=8 if fuse_b else int(rs.choice([0, 8])))
    try:
        engine.set_matrix_csr(M, K, rp, ci, v)
        out = C0.copy()
        engine.spmm(N, ALPHA, B, BETA, out)
        assert engine.last_kernel() == "spmm_csr_panel_v2", engine.last_kernel()
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    finally:
        _set(engine)


@pytest.mark.parametrize("N", [16, 24, 40, 48, 80])
def test_panel_v2_sixteen_column_tiles(engine, oracle, fem, N):
    """panel_v2 = 1: 16-column tiles (N = 16, the odd tile of N = 48, the 16-wide part of N = 24 / 40) also run the
    register-resident form, spmm_csr_panel_v2<1>: row entries loaded once, panel by LDS-DMA, C stored straight from
    the accumulators.  Same bits as the oracle."""
    M, rp, ci, v = fem
    K = M
    rs = np.random.RandomState(N)
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
    try:
        for cpl in (4, 0):
            for tpw in (0, 1, 2):
                _set(engine, fuse_b=0, panel_v2=1, cols_per_lane=cpl, tiles_per_wg=tpw)
                engine.set_matrix_csr(M, K, rp, ci, v)
                out = C0.copy()
                engine.spmm(N, ALPHA, B, BETA, out, rp_time=2)
                assert engine.last_kernel() == "spmm_csr_panel_v2", engine.last_kernel()
                assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (N, cpl, tpw)
    finally:
        _set(engine)


def _short_row_matrix(rs, M, max_len, spread):
    """Rows of 1 .. max_len entries with columns within +-spread of the diagonal (heavy B-row reuse inside a 64-row block)."""
    rp = np.zeros(M + 1, dtype=np.int32)
    cols = []
    for r in range(M):
        n = int(rs.randint(1, max_len + 1))
        lo, hi = max(0, r - spread), min(M, r + spread + 1)
        cols.append(np.sort(rs.choice(np.arange(lo, hi), size=min(n, hi - lo), replace=False)).astype(np.int32))
        rp[r + 1] = rp[r] + len(cols[-1])
    ci = np.concatenate(cols)
    return rp, ci, rs.uniform(-1, 1, len(ci)).astype(np.float32)


@pytest.mark.parametrize("max_len,spread", [(20, 60), (42, 60), (48, 40), (60, 200), (100, 100)])
def test_small_matrix_forms_of_panel_v2(engine, oracle, max_len, spread):
    """Small matrices with short rows staged from column-major B (fuse_b): the register-resident batches follow the LONGEST row
    (2, 3, 4 or 6 batches: every row register-resident when that costs at most one more batch), and the dictionary capacity of
    the launch follows the plan (5 x 64 rows when no block needs more; the full 9 x 64 otherwise).  small_v2 = 0 keeps the
    full-capacity form; both, and the rp_time loop (panels repacked once, LDS-DMA form), give the oracle's bits."""
    M = K = 1500
    rs = np.random.RandomState(max_len * 7 + spread)
    rp, ci, v = _short_row_matrix(rs, M, max_len, spread)
    for N in (16, 32):
        B = rs.uniform(-1, 1, K * N).astype(np.float32)
        C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
        want = C0.copy()
        oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
        try:
            for small in (1, 0):
                _set(engine, fuse_b=1, panel_v2=-1)
                engine.set_option("small_v2", small)
                engine.set_matrix_csr(M, K, rp, ci, v)
                for rp_time in (1, 5):
                    out = C0.copy()
                    engine.spmm(N, ALPHA, B, BETA, out, rp_time=rp_time)
                    assert engine.last_kernel().startswith("spmm_csr_panel"), engine.last_kernel()
                    assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (N, small, rp_time, engine.last_kernel())
        finally:
            engine.set_option("small_v2", 1)
            _set(engine)


def test_panel_threshold_depends_on_n(engine, oracle, sx):
    """A 2-D 5-point stencil has 1.65 non-zeros per distinct column in a 64-row block: below the N <= 16 threshold (2.0, gather
    kernel), above the N >= 32 one (1.5, LDS panel; measured +14 % at N = 128).  One plan serves both (built for the lower
    threshold), so alternating N does not rebuild anything; both kernels give the oracle's bits."""
    from sextans_amd import api
    nx = ny = 150
    M = K = nx * ny
    rp, ci, v = api.gen_stencil2d_host(nx, ny, 5, 1, 3)
    for k, val in dict(lanes_per_row=0, kernel=0, panel_min_reuse_x100=200, panel_min_reuse_wide_x100=150, fuse_b=0, panel_v2=-1,
                       cols_per_lane=0, tiles_per_wg=0, split_rows=0, bucket_rows=-1, colwise_max_len=0).items():
        engine.set_option(k, val)                                    # (colwise_max_len = 0: not the lane-per-row kernel, test_colwise_gpu.py)
    engine.set_matrix_csr(M, K, rp, ci, v)
    rs = np.random.RandomState(11)
    built = None
    try:
        for N, kern in ((16, "spmm_csr_rowgroup"), (32, "spmm_csr_panel_v2"), (16, "spmm_csr_rowgroup"), (64, "spmm_csr_panel_v2")):
            B = rs.uniform(-1, 1, K * N).astype(np.float32)
            C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
            want = C0.copy()
            oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
            out = C0.copy()
            engine.spmm(N, ALPHA, B, BETA, out)
            assert engine.last_kernel() == kern, (N, engine.last_kernel())
            assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), N
            if built is None:
                built = engine.get_stat("plan_build_s")
            assert engine.get_stat("plan_build_s") == built          # one plan, never rebuilt
        engine.set_option("panel_min_reuse_wide_x100", 200)          # same threshold for every N: gather kernel at N = 32 too
        engine.set_option("row_cluster", 0)                          # (and no graph-clustered plan in its place: test_graph_cluster_gpu.py)
        out = C0.copy()
        engine.spmm(64, ALPHA, B, BETA, out)
        assert engine.last_kernel() == "spmm_csr_rowgroup"
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    finally:
        engine.set_option("panel_min_reuse_wide_x100", 150)
        engine.set_option("row_cluster", -1)
        engine.set_option("colwise_max_len", 6)
        _set(engine)
