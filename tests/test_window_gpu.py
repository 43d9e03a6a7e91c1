"""K-windowed accumulator-resident kernel (spmm_csr_window, engine option kernel=3; the reference's own
dataflow for inputs without locality: sextans.cpp:337,353-381 B window, :462-570 resident C) against the
oracle's cpu_spmm_CSR restatement, BIT-EXACT, through the C ABI."""
import hashlib

import numpy as np
import pytest

from util import ALPHA, BETA, NASA, bits_equal, random_csr

pytestmark = pytest.mark.gpu

DEFAULTS = dict(lanes_per_row=4, stage_a=1, xcd_remap=1, exact=1, kernel=3, panel_min_reuse_x100=400, fuse_b=1,
                split_rows=0, bucket_rows=0, window_rows=319, window_cols=65536, window_unroll=8)


def run(engine, M, K, rp, ci, v, N, alpha, B, beta, C0, rp_time=1, **opts):
    o = dict(DEFAULTS)
    o.update(opts)
    for k, val in o.items():
        engine.set_option(k, val)
    engine.set_matrix_csr(M, K, rp, ci, v)
    out = C0.copy()
    engine.spmm(N, alpha, B, beta, out, rp_time=rp_time)
    return out


@pytest.fixture(autouse=True)
def _restore(engine):
    yield
    for k, val in DEFAULTS.items():
        engine.set_option(k, val if k != "kernel" else 0)


@pytest.mark.parametrize("N", [8, 16, 24, 40])
@pytest.mark.parametrize("rows,cols,unroll", [(319, 65536, 8), (50, 128, 4), (7, 16, 8), (510, 1000, 4)])
def test_random_matrix_bit_exact_vs_oracle(engine, oracle, N, rows, cols, unroll):
    rs = np.random.RandomState(3000 + N + rows)
    M, K = 1237, 911
    rp, ci, v = random_csr(rs, M, K, 11, long_rows=2)
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
    out = run(engine, M, K, rp, ci, v, N, ALPHA, B, BETA, C0, window_rows=rows, window_cols=cols, window_unroll=unroll)
    assert engine.last_kernel() == "spmm_csr_window"
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))


def test_nasa4704_hashes_through_the_window_kernel(engine, sx):
    rp, ci, v, M, K, nnz = sx.read_suitsparse_matrix(NASA)
    want = {16: "988205f823683783aea5cd8c7eb0f88846b59bd957d3429ba63e2110bd3ad88f",
            128: "0a6b46a1581cfd04de887ebb3c815eebf7c93b9dca17e0010bb1fb869f51817e"}
    for N in (16, 128):
        out = run(engine, M, K, rp, ci, v, N, ALPHA, sx.init_dense_B(K, N), BETA, sx.init_dense_C(M, N), window_cols=512)
        assert engine.last_kernel() == "spmm_csr_window"
        assert hashlib.sha256(out.tobytes()).hexdigest() == want[N], N


def test_many_windows_duplicates_special_values(engine, oracle):
    """Columns spread over hundreds of windows, duplicate columns inside rows, signed zeros, a denormal and
    an infinity: accumulation order and rounding must still be cpu_spmm_CSR's."""
    rs = np.random.RandomState(8)
    M, K, N = 900, 70000, 16
    lens = rs.poisson(25, M)
    lens[::50] = 0
    lens[13] = 3000
    rp = np.zeros(M + 1, np.int32); rp[1:] = np.cumsum(lens)
    ci = np.concatenate([np.sort(rs.randint(0, K, l)) for l in lens]).astype(np.int32)     # duplicates allowed
    v = rs.uniform(-1, 1, rp[-1]).astype(np.float32)
    v[rs.randint(0, rp[-1], 3)] = [0.0, -0.0, np.float32(1e-40)]
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    B[rs.randint(0, K * N, 2)] = [np.float32(-0.0), np.float32(np.inf)]
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    for alpha, beta in ((ALPHA, BETA), (1.0, 0.0), (0.0, 1.0)):
        want = C0.copy()
        oracle.spmm(M, N, K, np.float32(alpha), rp, ci, v, B, np.float32(beta), want)
        for cols in (256, 65536):
            out = run(engine, M, K, rp, ci, v, N, alpha, B, beta, C0, window_cols=cols, rp_time=2)
            assert bits_equal(out, want), (alpha, beta, cols)


def test_degenerate_shapes(engine, oracle):
    N = 8
    for (M, K) in [(1, 1), (1, 300), (65, 1), (319, 64), (320, 33), (638, 5)]:
        rs = np.random.RandomState(M * 1000 + K)
        rp, ci, v = random_csr(rs, M, K, min(K, 5), empty_frac=0.0)
        B = rs.uniform(-1, 1, K * N).astype(np.float32)
        C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
        want = C0.copy()
        oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
        out = run(engine, M, K, rp, ci, v, N, ALPHA, B, BETA, C0)
        assert engine.last_kernel() == ("spmm_csr_window" if rp[-1] > 0 else "spmm_csr_rowgroup")
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (M, K)
    # an all-empty matrix never reaches the window kernel (nothing to stream): plain epilogue path
    M, K = 70, 5
    rp = np.zeros(M + 1, np.int32)
    C0 = np.arange(M * N, dtype=np.float32)
    out = run(engine, M, K, rp, np.zeros(0, np.int32), np.zeros(0, np.float32), N, ALPHA, np.ones(K * N, np.float32), BETA, C0)
    assert np.array_equal(out, (np.float32(ALPHA) * np.float32(0)) + np.float32(BETA) * C0)


def test_row_ranges_and_strides_device_resident(engine, oracle):
    """sextans_spmm_device_rows on wavefront-aligned ranges runs the window kernel, on unaligned ranges the
    gather kernel; B panels of the two layouts are never mixed up when the caller asks for panel reuse."""
    import torch
    rs = np.random.RandomState(61)
    M, K, N, RW = 1000, 800, 16, 100
    rp, ci, v = random_csr(rs, M, K, 9)
    ldb = K + 5
    Bfull = rs.uniform(-1, 1, ldb * N).astype(np.float32)
    Bc = np.ascontiguousarray(Bfull.reshape(N, ldb)[:, :K]).reshape(-1)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp, ci, v, Bc, BETA, want)
    for k, val in dict(DEFAULTS, window_rows=RW, window_cols=128).items():
        engine.set_option(k, val)
    engine.set_matrix_csr(M, K, rp, ci, v)
    dB = torch.from_numpy(Bfull).cuda(); dCin = torch.from_numpy(C0).cuda()
    st = torch.cuda.current_stream().cuda_stream
    kernels = []
    out = torch.full((M * N,), float("nan"), device="cuda")
    cuts = [0, 300, 350, 700, 1000]                       # 350 is not a multiple of 100
    for i in range(len(cuts) - 1):
        c0, c1 = cuts[i], cuts[i + 1]
        slab = torch.full(((c1 - c0) * N,), float("nan"), device="cuda")
        engine.spmm_device_rows(N, ALPHA, dB.data_ptr(), ldb, BETA, dCin.data_ptr() + 4 * c0, M, slab.data_ptr(), c1 - c0,
                                c0, c1, reuse_b_panels=i > 0, stream=st)
        kernels.append(engine.last_kernel())
        out.view(N, M)[:, c0:c1] = slab.view(N, c1 - c0)
    torch.cuda.synchronize()
    assert kernels == ["spmm_csr_window", "spmm_csr_rowgroup", "spmm_csr_rowgroup", "spmm_csr_window"]
    assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_auto_dispatch_rules(engine, sx):
    """kernel=0: small B (fits the L2s) and matrices with B-row reuse never take the window kernel; a skewed
    matrix is rejected even when forced candidates exist."""
    rp, ci, v, M, K, nnz = sx.read_suitsparse_matrix(NASA)
    for k, val in dict(DEFAULTS, kernel=0).items():
        engine.set_option(k, val)
    engine.set_matrix_csr(M, K, rp, ci, v)
    C = sx.init_dense_C(M, 16)
    engine.spmm(16, ALPHA, sx.init_dense_B(K, 16), BETA, C)
    assert engine.last_kernel().startswith("spmm_csr_panel")
    rs = np.random.RandomState(5)
    rp, ci, v = random_csr(rs, 3000, 2000, 10)
    engine.set_matrix_csr(3000, 2000, rp, ci, v)
    C = np.zeros(3000 * 16, np.float32)
    engine.spmm(16, ALPHA, np.ones(2000 * 16, np.float32), BETA, C)
    assert engine.last_kernel() == "spmm_csr_rowgroup"
    assert engine.get_stat("window_state") == 0          # never evaluated: B is tiny


def test_non_exact_variant_within_tolerance(engine, oracle):
    rs = np.random.RandomState(21)
    M, K, N = 800, 700, 16
    rp, ci, v = random_csr(rs, M, K, 30)
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
    out = run(engine, M, K, rp, ci, v, N, ALPHA, B, BETA, C0, exact=0)
    rows = np.repeat(np.arange(M), np.diff(rp))
    bound = np.zeros(M * N, np.float64)
    Bm = np.abs(B.reshape(N, K))
    for n in range(N):
        bound[n * M:(n + 1) * M] = np.bincount(rows, weights=np.abs(v) * Bm[n, ci], minlength=M)
    bound = 1e-4 * (abs(float(ALPHA)) * bound + np.abs(float(BETA) * C0))
    assert np.all(np.abs(out.astype(np.float64) - want) <= bound + 1e-30)
