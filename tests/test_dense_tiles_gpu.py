"""north_star: "a blocked-ELL / row-bucketed variant that feeds MFMA only where a tile is actually dense".  The CSR
dispatcher finds the 32x32 tiles whose fill reaches a threshold; by default it only reports them (fp32 everywhere,
bit-identical); with mfma_dense_tiles = 1 the caller opts into bf16 for those tiles: they run on the matrix cores,
the remainder on the fp32 CSR kernels, in ONE call.  No analogue in the reference (parity unpinned by it): the checker
is a float64 evaluation of exactly the mixed-precision product, with the blocked-ELL tolerance."""
import numpy as np
import pytest

from util import ALPHA, BETA

pytestmark = pytest.mark.gpu


def bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


def block_diagonal_plus_noise(rs, M, K, fill_off=0.6, hub_row=None):
    """Fully dense 32x32 diagonal blocks, a second band of ~60 %-filled tiles, and 4 random non-zeros per row."""
    rows, cols = [], []
    for br in range(min(M, K) // 32):
        r, c = np.meshgrid(np.arange(32), np.arange(32), indexing="ij")
        rows.append(br * 32 + r.ravel()); cols.append(br * 32 + c.ravel())
        bc = (br * 7 + 3) % (K // 32)
        if bc != br:
            m = rs.rand(32, 32) < fill_off
            rows.append(br * 32 + r[m]); cols.append(bc * 32 + c[m])
    nr = np.repeat(np.arange(M), 4)
    rows.append(nr); cols.append(rs.randint(0, K, len(nr)))
    if hub_row is not None:                                          # one hub row: the piece path runs next to both kernels
        hc = rs.choice(K, size=min(K, 1500), replace=False)
        rows.append(np.full(len(hc), hub_row)); cols.append(hc)
    r = np.concatenate(rows); c = np.concatenate(cols)
    key = np.unique(r.astype(np.int64) * K + c)                      # distinct, sorted by (row, col)
    r, c = (key // K).astype(np.int32), (key % K).astype(np.int32)
    rp = np.zeros(M + 1, np.int32)
    rp[1:] = np.cumsum(np.bincount(r, minlength=M))
    return rp, c, rs.uniform(-1, 1, len(c)).astype(np.float32)


def dense_mask(M, K, rp, ci, thr):
    rows = np.repeat(np.arange(M), np.diff(rp))
    tile = (rows // 32).astype(np.int64) * ((K + 31) // 32) + ci // 32
    cnt = np.bincount(tile, minlength=((M + 31) // 32) * ((K + 31) // 32))
    return (cnt[tile] >= thr) & (rows < (M // 32) * 32)


@pytest.mark.parametrize("M,K,N", [(2048 + 17, 2048 + 40, 64), (1024, 4096, 32), (640, 640, 96), (1024 + 5, 1024, 256)])
def test_block_diagonal_plus_noise_runs_both_kernels(engine, M, K, N):
    rs = np.random.RandomState(M + N)
    rp, ci, v = block_diagonal_plus_noise(rs, M, K, hub_row=M - 3 if M > 2000 else None)
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    for k, val in dict(kernel=0, lanes_per_row=4, exact=1, split_rows=-1, bucket_rows=-1, mfma_dense_tiles=0,
                       dense_tile_fill_x100=50).items():      # split_rows = -1: the hub row is re-associated (opt-in)
        engine.set_option(k, val)
    engine.set_matrix_csr(M, K, rp, ci, v)
    dm = dense_mask(M, K, rp, ci, 512)
    # default: report only -- fp32 everywhere
    plain = C0.copy()
    engine.spmm(N, ALPHA, B, BETA, plain)
    assert "dense_tiles" not in engine.last_kernel() and engine.get_stat("dense_tiles_on_mfma") == 0
    assert engine.get_stat("dense_tiles") >= 2 * (min(M, K) // 32) - 2
    # opt in: dense tiles on MFMA (bf16), the rest on the CSR kernels, one call
    engine.set_option("mfma_dense_tiles", 1)
    out = C0.copy()
    engine.spmm(N, ALPHA, B, BETA, out, rp_time=2)
    assert engine.last_kernel().endswith("+dense_tiles_mfma") and engine.get_stat("dense_tiles_on_mfma") == 1
    assert ("+hub_pieces" in engine.last_kernel()) == (M > 2000) and (engine.get_stat("reassociated_rows") == 1) == (M > 2000)
    # the mixed-precision product in float64
    rows = np.repeat(np.arange(M), np.diff(rp))
    Bm = B.reshape(N, K).T.astype(np.float64)
    Bb = bf16_round(B).reshape(N, K).T.astype(np.float64)
    a_used = np.where(dm, bf16_round(v), v).astype(np.float64)
    want = np.zeros((M, N)); asum = np.zeros((M, N))
    for sel, Bx in ((dm, Bb), (~dm, Bm)):
        np.add.at(want, rows[sel], a_used[sel, None] * Bx[ci[sel]])
        np.add.at(asum, rows[sel], np.abs(a_used[sel, None] * Bx[ci[sel]]))
    Cm = C0.reshape(N, M).T.astype(np.float64)
    want = float(ALPHA) * want + float(BETA) * Cm
    got = out.reshape(N, M).T.astype(np.float64)
    tol = 4e-6 * asum * abs(float(ALPHA)) + 2e-6 * np.abs(float(BETA) * Cm) + 1e-30
    assert np.all(np.abs(got - want) <= tol), float(np.max(np.abs(got - want) / tol))
    # and it differs from the all-fp32 result by no more than bf16 rounding of the dense part allows
    d = np.abs(got - plain.reshape(N, M).T.astype(np.float64))
    assert d.max() > 0 and np.all(d <= 2.0 ** -7 * asum * abs(float(ALPHA)) + tol)
    # errors: N not a multiple of 32, row ranges
    from sextans_amd import api
    with pytest.raises(api.SextansError):
        engine.spmm(16, ALPHA, B[:K * 16], BETA, C0[:M * 16].copy())
    engine.set_option("mfma_dense_tiles", 0)
    again = C0.copy()
    engine.spmm(N, ALPHA, B, BETA, again)                          # back to fp32: same bits as before
    assert np.array_equal(again.view(np.uint32), plain.view(np.uint32))


def test_no_dense_tiles_nothing_changes(engine, oracle):
    from util import random_csr
    rs = np.random.RandomState(2)
    M, K, N = 3000, 3000, 32
    rp, ci, v = random_csr(rs, M, K, 12)
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
    for k, val in dict(kernel=0, mfma_dense_tiles=1, split_rows=0, bucket_rows=0).items():
        engine.set_option(k, val)
    try:
        engine.set_matrix_csr(M, K, rp, ci, v)
        out = C0.copy()
        engine.spmm(N, ALPHA, B, BETA, out)
        assert engine.get_stat("dense_tiles") == 0 and "dense" not in engine.last_kernel()
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    finally:
        engine.set_option("mfma_dense_tiles", 0)


_CASES = int(__import__("os").environ.get("SEXTANS_DENSE_CASES", "6"))     # soak: SEXTANS_DENSE_CASES=300


@pytest.mark.parametrize("seed", range(_CASES))
def test_random_shapes_thresholds_kernels(engine, seed):
    """Random sizes (incl. M, K not multiples of 32), fill thresholds, main kernels and an optional hub row."""
    rs = np.random.RandomState(500 + seed)
    M, K = int(rs.choice([96, 333, 1000, 1700])), int(rs.choice([64, 257, 1024, 3000]))
    N = int(rs.choice([32, 64, 96]))
    hub = int(rs.randint(0, M)) if rs.rand() < 0.5 and K >= 1024 else None
    rp, ci, v = block_diagonal_plus_noise(rs, M, K, fill_off=float(rs.choice([0.3, 0.6, 0.9])), hub_row=hub)
    fill = int(rs.choice([25, 50, 75]))
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    for k, val in dict(kernel=int(rs.choice([0, 1, 2, 3])), lanes_per_row=int(rs.choice([0, 4, 8])), exact=1, split_rows=-1,
                       bucket_rows=int(rs.choice([-1, 0])), mfma_dense_tiles=1, dense_tile_fill_x100=fill).items():
        engine.set_option(k, val)
    try:
        engine.set_matrix_csr(M, K, rp, ci, v)
        out = C0.copy()
        engine.spmm(N, ALPHA, B, BETA, out)
        thr = (fill * 1024 + 99) // 100
        dm = dense_mask(M, K, rp, ci, thr)
        # (the tiles are cut out of the matrix as set, BEFORE long rows are bucketed or split)
        assert (engine.get_stat("dense_tiles_on_mfma") == 1) == bool(dm.any())
        rows = np.repeat(np.arange(M), np.diff(rp))
        Bm = B.reshape(N, K).T.astype(np.float64)
        Bb = bf16_round(B).reshape(N, K).T.astype(np.float64)
        a_used = np.where(dm, bf16_round(v), v).astype(np.float64)
        want = np.zeros((M, N)); asum = np.zeros((M, N))
        for sel, Bx in ((dm, Bb), (~dm, Bm)):
            np.add.at(want, rows[sel], a_used[sel, None] * Bx[ci[sel]])
            np.add.at(asum, rows[sel], np.abs(a_used[sel, None] * Bx[ci[sel]]))
        Cm = C0.reshape(N, M).T.astype(np.float64)
        want = float(ALPHA) * want + float(BETA) * Cm
        got = out.reshape(N, M).T.astype(np.float64)
        tol = 4e-6 * asum * abs(float(ALPHA)) + 2e-6 * np.abs(float(BETA) * Cm) + 1e-30
        assert np.all(np.abs(got - want) <= tol), (float(np.max(np.abs(got - want) / tol)), engine.last_kernel())
    finally:
        for k, val in dict(kernel=0, lanes_per_row=0, mfma_dense_tiles=0, dense_tile_fill_x100=50, bucket_rows=-1).items():
            engine.set_option(k, val)
