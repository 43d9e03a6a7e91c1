"""spmm_csr_colwise (csrc/spmm_colwise_kernel.h): the short-row form -- one lane per row, B and C read / written where the caller left
them (column-major), no repack launch, no LDS.  Chosen automatically for matrices with a mean row length <= 6 whose consecutive rows
have neighbouring columns (5-point / 7-point stencils: measured 361 -> 259 us per step on a 4M-row 5-point stencil at N = 16, while a
9-point stencil already loses, 368 -> 407 us); "kernel" = 4 forces it for anything.  Per-row CSR order, rounded
products: bit-identical to cpu_spmm_CSR (sparse_helper.h:262-290).  What the reference does about short rows: packs them back to
back into each PE's list (sparse_helper.h:292-343)."""
import numpy as np
import pytest

from util import ALPHA, BETA, random_csr

pytestmark = pytest.mark.gpu

OPTS = dict(lanes_per_row=0, kernel=0, fuse_b=1, panel_v2=-1, split_rows=0, bucket_rows=-1, row_cluster=-1, exact=1, colwise_max_len=6,
            panel_min_reuse_x100=200, panel_min_reuse_wide_x100=150)


def _set(engine, **kw):
    d = dict(OPTS)
    d.update(kw)
    for k, v in d.items():
        engine.set_option(k, v)


def _check(engine, oracle, M, K, rp, ci, v, N, rs, alpha=ALPHA, beta=BETA, rp_time=1):
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm(M, N, K, alpha, rp, ci, v, B, beta, want)
    out = C0.copy()
    engine.spmm(N, alpha, B, beta, out, rp_time=rp_time)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (M, K, N, engine.last_kernel())
    return B, C0, want


@pytest.mark.parametrize("N", [8, 16, 24, 40, 128])
def test_short_row_matrices_take_the_colwise_kernel(engine, oracle, N):
    from sextans_amd import api
    rs = np.random.RandomState(N)
    cases = [("2-D 5-point", api.gen_stencil2d_host(140, 90, 5, 1, 3), 140 * 90),
             ("2-D 5-point, tall", api.gen_stencil2d_host(17, 900, 5, 1, 3), 17 * 900)]
    try:
        for name, (rp, ci, v), M in cases:
            _set(engine)
            engine.set_matrix_csr(M, M, rp, ci, v)
            for rp_time in (1, 4):
                _check(engine, oracle, M, M, rp, ci, v, N, rs, rp_time=rp_time)
                assert engine.last_kernel() == "spmm_csr_colwise", (name, engine.last_kernel(), engine.get_stat("row_coherence"))
            assert int(engine.get_stat("colwise")) == 1 and engine.get_stat("row_coherence") > 0.9
            _set(engine, colwise_max_len=0)                      # switched off: the other kernels, same bits
            _check(engine, oracle, M, M, rp, ci, v, N, rs)
            assert engine.last_kernel() != "spmm_csr_colwise"
    finally:
        _set(engine)


def test_not_chosen_without_locality_or_for_long_rows(engine, oracle):
    from sextans_amd import api
    rs = np.random.RandomState(2)
    try:
        _set(engine)
        M = 9000
        rp, ci, v = random_csr(rs, M, M, 8)                      # short rows, random columns: no coherence
        engine.set_matrix_csr(M, M, rp, ci, v)
        _check(engine, oracle, M, M, rp, ci, v, 16, rs)
        assert int(engine.get_stat("colwise")) == -1 and engine.last_kernel() != "spmm_csr_colwise"
        rp, ci, v = api.gen_fem3d_host(12, 11, 10, 3, 7)         # locality, but 81 entries per row
        M = 12 * 11 * 10 * 3
        engine.set_matrix_csr(M, M, rp, ci, v)
        _check(engine, oracle, M, M, rp, ci, v, 16, rs)
        assert int(engine.get_stat("colwise")) == -1 and engine.last_kernel() != "spmm_csr_colwise"
    finally:
        _set(engine)


def test_forced_on_ragged_matrices_row_ranges_and_options(engine, oracle):
    """kernel = 4 on matrices it would never pick: empty rows, rows of 0 .. 60 entries, rectangular shapes, one row / one column;
    alpha / beta special values; exact = 0 inside the stated bound; row-range calls write packed slabs."""
    import torch
    rs = np.random.RandomState(7)
    try:
        _set(engine, kernel=4)
        for M, K, mean in ((1, 1, 1), (1, 300, 40), (300, 1, 1), (777, 1234, 15), (5000, 4000, 3), (2049, 2049, 30)):
            rp, ci, v = random_csr(rs, M, K, mean, empty_frac=0.2)
            engine.set_matrix_csr(M, K, rp, ci, v)
            for N in (8, 16, 40):
                _check(engine, oracle, M, K, rp, ci, v, N, rs)
                assert engine.last_kernel() == "spmm_csr_colwise" or rp[-1] == 0      # (an empty matrix takes the generic path)
        M = K = 3000
        rp, ci, v = random_csr(rs, M, K, 9)
        engine.set_matrix_csr(M, K, rp, ci, v)
        for alpha, beta in ((np.float32(1), np.float32(0)), (np.float32(0), np.float32(1)), (np.float32(-1.5), np.float32(0.25))):
            _check(engine, oracle, M, K, rp, ci, v, 24, rs, alpha, beta)
        N = 24
        B, C0, want = _check(engine, oracle, M, K, rp, ci, v, N, rs)
        st = torch.cuda.current_stream().cuda_stream
        dB = torch.from_numpy(B).cuda(); dCin = torch.from_numpy(C0).cuda()
        got = torch.full((M * N,), float("nan"), device="cuda")
        cuts = [0, 1, 1700, M]
        for i in range(3):
            c0, c1 = cuts[i], cuts[i + 1]
            slab = torch.full(((c1 - c0) * N,), float("nan"), device="cuda")
            engine.spmm_device_rows(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr() + 4 * c0, M, slab.data_ptr(), c1 - c0, c0, c1,
                                    reuse_b_panels=i > 0, stream=st)
            got.view(N, M)[:, c0:c1] = slab.view(N, c1 - c0)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
        _set(engine, kernel=4, exact=0)
        out = C0.copy()
        engine.spmm(N, ALPHA, B, BETA, out)
        import scipy.sparse as sp
        A = sp.csr_matrix((np.abs(v), ci, rp), shape=(M, K))
        bound = 1e-4 * (abs(ALPHA) * (A @ np.abs(B.reshape(N, K).T)).T.reshape(-1) + np.abs(BETA * C0))
        assert np.all(np.abs(out.astype(np.float64) - want) <= bound + 1e-30)
    finally:
        _set(engine)


def test_long_rows_fall_back(engine, oracle):
    """Rows on the exact-chain / piece paths need the repacked B panels: a matrix that has them keeps the other kernels even when
    colwise is forced; same bits."""
    rs = np.random.RandomState(4)
    M = 6000
    rp, ci, v = random_csr(rs, M, M, 6, long_rows=0)
    lens = np.diff(rp).copy()
    rows = np.repeat(np.arange(M), lens)
    extra = np.sort(rs.choice(M, size=2500, replace=False))
    keep = rows != 123
    rows2 = np.concatenate([rows[keep], np.full(len(extra), 123)]); cols2 = np.concatenate([ci[keep], extra]); vals2 = np.concatenate([v[keep], rs.uniform(-1, 1, len(extra)).astype(np.float32)])
    o = np.lexsort((cols2, rows2))
    rows2, cols2, vals2 = rows2[o], cols2[o].astype(np.int32), vals2[o].astype(np.float32)
    rp2 = np.zeros(M + 1, np.int32); np.add.at(rp2, rows2 + 1, 1); rp2 = np.cumsum(rp2).astype(np.int32)
    try:
        _set(engine, kernel=4)
        engine.set_matrix_csr(M, M, rp2, cols2, vals2)
        _check(engine, oracle, M, M, rp2, cols2, vals2, 16, rs)
        assert engine.get_stat("piece_path_rows") >= 1 and engine.last_kernel() != "spmm_csr_colwise"
    finally:
        _set(engine)
