"""CPU tests of the K-windowed stream of A (window_plan.cpp; the analogue of the reference's per-(PE, window)
scheduling, sparse_helper.h:345-403): the stream must hold every non-zero exactly once, keep each row's CSR
order (that is what makes the kernel bit-identical to cpu_spmm_CSR), never place a row twice in a 32-entry
step, and walk K in window order.  Also a numpy model of the kernel's accumulation against the oracle."""
import numpy as np
import pytest

from util import ALPHA, BETA, random_csr


def decode(p, rp):
    """Stream -> per-row (col, val) sequences in stream order, checking the step invariants."""
    RW = p["rows_per_wave"]
    M = p["M"]
    cols = [[] for _ in range(M)]
    vals = [[] for _ in range(M)]
    for g in range(p["nwaves"]):
        s0, s1 = p["wave_step0"][g], p["wave_step0"][g + 1]
        assert (s1 - s0) % 24 == 0
        for s in range(s0, s1):
            row = p["row"][s * 32:(s + 1) * 32]
            live = row != RW
            assert np.all(row[live] < min(RW, M - g * RW))
            assert len(np.unique(row[live])) == live.sum(), "a row appears twice in one step"
            assert np.all(p["val"][s * 32:(s + 1) * 32][~live] == 0)
            for i in np.nonzero(live)[0]:
                r = g * RW + row[i]
                cols[r].append(p["col"][s * 32 + i])
                vals[r].append(p["val"][s * 32 + i])
    return cols, vals


@pytest.mark.parametrize("M,K,mean,RW,win", [(700, 5000, 9, 64, 512), (1000, 300, 20, 319, 65536), (97, 40000, 30, 7, 1024),
                                             (300, 2000, 3, 510, 100)])
def test_stream_holds_every_row_in_csr_order(sx, M, K, mean, RW, win):
    from sextans_amd import api
    rs = np.random.RandomState(M + K)
    rp, ci, v = random_csr(rs, M, K, mean, long_rows=2)
    p = api.window_pack_csr(M, K, rp, ci, v, RW, win)
    assert p["nwaves"] == (M + RW - 1) // RW and p["steps"] == p["wave_step0"][-1]
    assert p["padded_lower_bound"] <= p["steps"] * 32
    cols, vals = decode(p, rp)
    for r in range(M):
        assert np.array_equal(np.array(cols[r], np.int32), ci[rp[r]:rp[r + 1]]), r
        assert np.array_equal(np.array(vals[r], np.float32).view(np.uint32), v[rp[r]:rp[r + 1]].view(np.uint32))


def test_window_order_and_padding_overhead(sx):
    """Uniform columns (the config-4 shape, scaled down): windows ascend along each wavefront's stream up to
    the scheduler's small look-ahead, and padding stays within a few percent."""
    from sextans_amd import api
    M, K, RW, win = 4000, 1 << 16, 319, 2048
    rp, ci, v = api.gen_csr_host(M, K, 40.0, 4)
    p = api.window_pack_csr(M, K, rp, ci, v, RW, win)
    nnz = int(rp[-1])
    assert p["steps"] * 32 <= 1.08 * nnz
    for g in range(p["nwaves"]):
        s0, s1 = p["wave_step0"][g], p["wave_step0"][g + 1]
        live = p["row"][s0 * 32:s1 * 32] != RW
        w = (p["col"][s0 * 32:s1 * 32] // win)[live]
        # entries are emitted in window order except for entries parked for a few steps
        assert np.all(np.diff(w.astype(np.int64)) >= -1)
        step_w = [w_[l_] for w_, l_ in ((p["col"][s * 32:(s + 1) * 32] // win, p["row"][s * 32:(s + 1) * 32] != RW)
                                        for s in range(s0, s1)) if l_.any()]
        assert all(a.max() - a.min() <= 1 for a in step_w)


def test_skewed_rows_are_rejected_by_the_estimate(sx):
    from sextans_amd import api
    rs = np.random.RandomState(3)
    M, K = 640, 50000
    lens = rs.poisson(5, M)
    lens[17] = 20000                                   # hub row: one entry per step
    rp = np.zeros(M + 1, np.int32); rp[1:] = np.cumsum(lens)
    ci = np.concatenate([np.sort(rs.choice(K, size=l, replace=False)) for l in lens]).astype(np.int32)
    v = rs.uniform(-1, 1, rp[-1]).astype(np.float32)
    p = api.window_pack_csr(M, K, rp, ci, v, 319, 65536)
    assert p["padded_lower_bound"] > 1.3 * rp[-1]      # what the engine's dispatcher tests
    assert p["steps"] * 32 >= p["padded_lower_bound"]
    cols, _ = decode(p, rp)
    assert np.array_equal(np.array(cols[17], np.int32), ci[rp[17]:rp[18]])


def test_numpy_model_of_the_kernel_matches_oracle(sx, oracle):
    """Accumulate exactly as spmm_csr_window does (per step: acc[row] += val * B[col], products rounded, in
    stream order) and compare with cpu_spmm_CSR bit for bit."""
    from sextans_amd import api
    rs = np.random.RandomState(11)
    M, K, N, RW = 500, 3000, 8, 100
    rp, ci, v = random_csr(rs, M, K, 25, long_rows=1)
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
    p = api.window_pack_csr(M, K, rp, ci, v, RW, 256)
    Bm = B.reshape(N, K).T.copy()                      # K x N
    acc = np.zeros((p["nwaves"], RW + 1, N), np.float32)
    for g in range(p["nwaves"]):
        for s in range(p["wave_step0"][g], p["wave_step0"][g + 1]):
            sl = slice(s * 32, (s + 1) * 32)
            prod = (p["val"][sl, None] * Bm[p["col"][sl]]).astype(np.float32)
            acc[g, p["row"][sl]] = (acc[g, p["row"][sl]] + prod).astype(np.float32)   # rows are distinct in a step
    got = np.empty((N, M), np.float32)
    for r in range(M):
        a = acc[r // RW, r % RW]
        got[:, r] = (np.float32(ALPHA) * a).astype(np.float32) + (np.float32(BETA) * C0.reshape(N, M)[:, r]).astype(np.float32)
    assert np.array_equal(got.reshape(-1).view(np.uint32), want.view(np.uint32))


def test_limits(sx):
    from sextans_amd import api
    rp = np.array([0, 1], np.int32)
    with pytest.raises(api.SextansError):
        api.window_pack_csr(1, (1 << 23) + 1, rp, np.array([5], np.int32), np.ones(1, np.float32))
    with pytest.raises(api.SextansError):
        api.window_pack_csr(1, 10, rp, np.array([5], np.int32), np.ones(1, np.float32), rows_per_wave=511)
    p = api.window_pack_csr(0, 10, np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32))
    assert p["steps"] == 0 and p["nwaves"] == 0
