"""Synthetic-input generators of the measurement harness (BASELINE config 4 shape)."""
import numpy as np
import pytest


def test_host_generator_properties(sx):
    from sextans_amd import api
    M, K = 20000, 5000
    rp, ci, v = api.gen_csr_host(M, K, 40.0, 4)
    lens = np.diff(rp)
    assert rp[0] == 0 and rp[-1] == len(ci) == len(v)
    assert abs(lens.mean() - 40.0) < 0.3 and abs(lens.var() - 40.0) < 2.0      # Poisson(40)
    assert ci.min() >= 0 and ci.max() < K
    inner = np.ones(len(ci), bool)
    inner[rp[:-1][lens > 0]] = False
    assert np.all(np.diff(ci)[inner[1:]] > 0)                                   # sorted, distinct
    assert v.min() >= -1.0 and v.max() < 1.0 and abs(v.mean()) < 0.01
    # any row range reproduces the same rows (counter-based)
    rp2, ci2, v2 = api.gen_csr_host(M, K, 40.0, 4, 777, 1234)
    assert np.array_equal(ci2, ci[rp[777]:rp[1234]]) and np.array_equal(v2, v[rp[777]:rp[1234]])
    assert np.array_equal(rp2, rp[777:1235] - rp[777])
    # different seed, different matrix
    assert not np.array_equal(api.gen_csr_host(M, K, 40.0, 5)[1][:100], ci[:100])
    # tiny K forces the without-replacement fix-up path: rows stay strictly increasing and < K
    rp3, ci3, _ = api.gen_csr_host(500, 48, 40.0, 9)
    for r in range(500):
        seg = ci3[rp3[r]:rp3[r + 1]]
        assert np.all(np.diff(seg) > 0) and (len(seg) == 0 or (seg[0] >= 0 and seg[-1] < 48))
    # banded and FEM-like kinds
    rp4, ci4, _ = api.gen_csr_host(3000, 3000, 40.0, 4, bandwidth=200)
    rows = np.repeat(np.arange(3000), np.diff(rp4))
    assert np.abs(ci4 - rows).max() <= 200 and ci4.min() >= 0 and ci4.max() < 3000
    frp, fci, fv = api.gen_fem3d_host(35, 19, 7, 3, 2)          # the pcrystk02 stand-in
    assert len(frp) - 1 == 13965 and frp[-1] == 968715 and np.diff(frp).max() == 81
    assert all(np.all(np.diff(fci[frp[r]:frp[r + 1]]) > 0) for r in range(0, 13965, 131))
    assert fci.max() == 13964 and np.array_equal(fci[frp[0]:frp[1]], fci[frp[1]:frp[2]])   # dof rows share columns
    u = api.gen_uniform_host(100000, 3)
    assert u.min() >= -1 and u.max() < 1 and abs(u.mean()) < 0.01 and abs(u.var() - 1 / 3) < 0.01


@pytest.mark.gpu
def test_device_generator_bit_identical_to_host(sx, engine):
    import torch
    from sextans_amd import api
    M, K = 30000, 4_000_000
    p, i, v, nnz = api.gen_csr_device(0, M, K, 40.0, 4, 100, 25000)
    try:
        hp, hi, hv = api.gen_csr_host(M, K, 40.0, 4, 100, 25000)
        assert nnz == len(hi)
        import ctypes as C
        def pull(ptr, n, dt):
            t = torch.empty(n, dtype=dt, device="cuda")
            torch.cuda.synchronize()
            hip = C.CDLL("libamdhip64.so.7")
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            assert hip.hipMemcpy(t.data_ptr(), ptr, n * 4, 3) == 0
            return t.cpu().numpy()
        assert np.array_equal(pull(p, len(hp), torch.int32), hp)
        assert np.array_equal(pull(i, nnz, torch.int32), hi)
        assert np.array_equal(pull(v, nnz, torch.float32).view(np.uint32), hv.view(np.uint32))
    finally:
        for q in (p, i, v):
            api.device_free(0, q)
    p, i, v, nnz = api.gen_fem3d_device(0, 35, 19, 7, 3, 2, 50, 9000)
    try:
        hp, hi, hv = api.gen_fem3d_host(35, 19, 7, 3, 2, 50, 9000)
        assert nnz == len(hi)
        import ctypes as C
        hip = C.CDLL("libamdhip64.so.7")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        ti = torch.empty(nnz, dtype=torch.int32, device="cuda")
        tv = torch.empty(nnz, dtype=torch.float32, device="cuda")
        assert hip.hipMemcpy(ti.data_ptr(), i, nnz * 4, 3) == 0 and hip.hipMemcpy(tv.data_ptr(), v, nnz * 4, 3) == 0
        assert np.array_equal(ti.cpu().numpy(), hi) and np.array_equal(tv.cpu().numpy().view(np.uint32), hv.view(np.uint32))
    finally:
        for q in (p, i, v):
            api.device_free(0, q)
    t = torch.empty(100000, device="cuda")
    api.gen_uniform_device(0, t.data_ptr(), 100000, 3, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy().view(np.uint32), api.gen_uniform_host(100000, 3).view(np.uint32))


def test_powerlaw_generator_properties(sx):
    from sextans_amd import api
    M = K = 200_000
    rp, ci, v = api.gen_powerlaw_host(M, K, 6, 120, 100_000, 7)
    lens = np.diff(rp)
    assert lens.min() >= 6 and lens.max() <= 100_000 and lens.max() > 20_000     # a few hubs ...
    assert np.median(lens) < 12 and 15 < lens.mean() < 60                           # ... over a mass of short rows
    # tail: P(len >= x) = (xmin / x)^1.2
    for x in (12, 48, 192):
        assert abs((lens >= x).mean() - (6 / x) ** 1.2) < 0.15 * (6 / x) ** 1.2 + 2e-4
    r = int(np.argmax(lens))
    seg = ci[rp[r]:rp[r + 1]]
    assert np.all(np.diff(seg) > 0) and seg[0] >= 0 and seg[-1] < K                # strata: distinct, ascending
    assert v.min() >= -1.0 and v.max() < 1.0
    rp2, ci2, v2 = api.gen_powerlaw_host(M, K, 6, 120, 100_000, 7, 5000, 5600)     # counter-based: any row range
    assert np.array_equal(ci2, ci[rp[5000]:rp[5600]]) and np.array_equal(v2, v[rp[5000]:rp[5600]])
    rp3, ci3, _ = api.gen_powerlaw_host(300, 50, 6, 120, 100_000, 1)               # max_len clipped to K
    assert np.diff(rp3).max() <= 50 and all(np.all(np.diff(ci3[rp3[i]:rp3[i + 1]]) > 0) for i in range(300))


@pytest.mark.gpu
def test_powerlaw_device_generator_bit_identical_to_host(sx, engine):
    import ctypes as C
    import torch
    from sextans_amd import api
    M = K = 300_000
    p, i, v, nnz = api.gen_powerlaw_device(0, M, K, 5, 130, 200_000, 9, 1000, 250_000)
    try:
        hp, hi, hv = api.gen_powerlaw_host(M, K, 5, 130, 200_000, 9, 1000, 250_000)
        assert nnz == len(hi)
        hip = C.CDLL("libamdhip64.so.7")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        tp = torch.empty(len(hp), dtype=torch.int32, device="cuda")
        ti = torch.empty(nnz, dtype=torch.int32, device="cuda")
        tv = torch.empty(nnz, dtype=torch.float32, device="cuda")
        assert hip.hipMemcpy(tp.data_ptr(), p, len(hp) * 4, 3) == 0
        assert hip.hipMemcpy(ti.data_ptr(), i, nnz * 4, 3) == 0 and hip.hipMemcpy(tv.data_ptr(), v, nnz * 4, 3) == 0
        assert np.array_equal(tp.cpu().numpy(), hp) and np.array_equal(ti.cpu().numpy(), hi)
        assert np.array_equal(tv.cpu().numpy().view(np.uint32), hv.view(np.uint32))
    finally:
        for q in (p, i, v):
            api.device_free(0, q)


def test_stencil2d_and_kkt_host_generators(sx):
    """Round 3: two more SuiteSparse-like classes for the sweep -- 2-D 5/9-point grid stencils and the KKT / arrow block
    structure (short rows without locality between blocks + a few very long border rows)."""
    import scipy.sparse as sp
    from sextans_amd import api
    nx, ny = 17, 11
    for points, dof in ((5, 1), (9, 1), (9, 2), (5, 3)):
        rp, ci, v = api.gen_stencil2d_host(nx, ny, points, dof, 3)
        M = nx * ny * dof
        assert len(rp) == M + 1 and rp[-1] == len(ci) == len(v) and np.diff(rp).max() == points * dof
        A = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(M, M))
        assert (A != A.T).nnz == 0                                          # structurally symmetric
        assert all(np.all(np.diff(ci[rp[r]:rp[r + 1]]) > 0) for r in range(M))
        interior = (nx - 2) * (ny - 2)
        assert (np.diff(rp) == points * dof).sum() == interior * dof
        rp2, ci2, v2 = api.gen_stencil2d_host(nx, ny, points, dof, 3, 40, 97)   # any row range, same rows
        assert np.array_equal(ci2, ci[rp[40]:rp[97]]) and np.array_equal(v2.view(np.uint32), v[rp[40]:rp[97]].view(np.uint32))
    n, arrow = 1000, 4
    rp, ci, v = api.gen_kkt_host(n, arrow, 3)
    M = api.kkt_rows(n, arrow)
    m = n // 2
    assert M == n + m + arrow and len(rp) == M + 1
    A = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(M, M)).toarray()
    assert np.array_equal(A[:n, n:n + m], A[n:n + m, :n].T)                # [[H, A^T], [A, 0]]
    assert not A[n:n + m, n:n + m].any()
    assert A[:, n + m:].all()                                              # every row carries the border columns
    assert np.array_equal(np.diff(rp)[n + m:], [len(range(k % 16, n + m, 16)) + arrow for k in range(arrow)])   # long border rows
    assert all(np.all(np.diff(ci[rp[r]:rp[r + 1]]) > 0) for r in range(M))
    assert np.abs(np.nonzero(A[:n, :n])[0] - np.nonzero(A[:n, :n])[1]).max() == 2                    # pentadiagonal H


@pytest.mark.gpu
@pytest.mark.parametrize("spec", ["stencil2d5", "stencil2d9x2", "kkt"])
def test_new_classes_device_generators_and_spmm(sx, engine, oracle, spec):
    """Device generator == host generator bit for bit; SpMM (default options: strict order, long rows bucketed) is
    bit-identical to the oracle -- the KKT class has border rows of thousands of entries next to 7-entry rows."""
    import ctypes as C
    import torch
    from sextans_amd import api
    from util import ALPHA, BETA
    if spec == "kkt":
        n, arrow = 40000, 3
        M = K = api.kkt_rows(n, arrow)
        host = api.gen_kkt_host(n, arrow, 3)
        dev = api.gen_kkt_device(0, n, arrow, 3)
    else:
        nx, ny, points, dof = (150, 120, 5, 1) if spec == "stencil2d5" else (90, 70, 9, 2)
        M = K = nx * ny * dof
        host = api.gen_stencil2d_host(nx, ny, points, dof, 3)
        dev = api.gen_stencil2d_device(0, nx, ny, points, dof, 3)
    hp, hi, hv = host
    p, i, v, nnz = dev
    try:
        hip = C.CDLL("libamdhip64.so.7")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        def pull(ptr, n_, dt):
            t = torch.empty(n_, dtype=dt, device="cuda")
            assert hip.hipMemcpy(t.data_ptr(), ptr, n_ * 4, 3) == 0
            return t.cpu().numpy()
        assert nnz == len(hi) and np.array_equal(pull(p, M + 1, torch.int32), hp) and np.array_equal(pull(i, nnz, torch.int32), hi)
        assert np.array_equal(pull(v, nnz, torch.float32).view(np.uint32), hv.view(np.uint32))
        for k, val in dict(kernel=0, lanes_per_row=0, exact=1, split_rows=0, bucket_rows=-1, panel_v2=-1, cols_per_lane=0).items():
            engine.set_option(k, val)
        engine.set_matrix_csr_device(M, K, nnz, p, i, v)
        rs = np.random.RandomState(7)
        for N in (16, 64):
            B = rs.uniform(-1, 1, K * N).astype(np.float32)
            C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
            want = C0.copy()
            oracle.spmm(M, N, K, ALPHA, hp, hi, hv, B, BETA, want)
            out = C0.copy()
            dB = torch.from_numpy(B).cuda(); dC = torch.from_numpy(out).cuda()
            engine.spmm_device(N, ALPHA, dB.data_ptr(), K, BETA, dC.data_ptr(), dC.data_ptr(), M, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert np.array_equal(dC.cpu().numpy().view(np.uint32), want.view(np.uint32)), (spec, N, engine.last_kernel())
            if spec == "kkt":
                assert engine.last_kernel().endswith("+hub_pieces") and engine.get_stat("reassociated_rows") == 0
    finally:
        engine.set_matrix_csr(1, 1, np.array([0, 0], np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32))
        for q in (p, i, v):
            api.device_free(0, q)
