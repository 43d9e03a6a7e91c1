"""torch.sparse_csr front end against the oracle, bit for bit."""
import numpy as np
import pytest

from util import random_csr

pytestmark = pytest.mark.gpu


def test_sparse_csr_op_matches_oracle(sx, oracle):
    import torch
    from sextans_amd.torch_op import spmm
    rs = np.random.RandomState(8)
    M, K, N = 700, 600, 20                      # N is not a multiple of 8: padded internally
    rp, ci, v = random_csr(rs, M, K, 10)
    A = torch.sparse_csr_tensor(torch.from_numpy(rp.astype(np.int64)), torch.from_numpy(ci.astype(np.int64)),
                                torch.from_numpy(v), size=(M, K)).cuda()
    B = rs.uniform(-1, 1, (K, N)).astype(np.float32)
    C0 = rs.uniform(-1, 1, (M, N)).astype(np.float32)
    alpha, beta = np.float32(0.85), np.float32(-2.06)
    Np = 24
    Bp = np.zeros((K, Np), np.float32); Bp[:, :N] = B
    Cp = np.zeros((M, Np), np.float32); Cp[:, :N] = C0
    want = np.ascontiguousarray(Cp.T).reshape(-1)
    oracle.spmm(M, Np, K, alpha, rp, ci, v, np.ascontiguousarray(Bp.T).reshape(-1), beta, want)
    want = want.reshape(Np, M).T[:, :N]
    got = spmm(A, torch.from_numpy(B).cuda(), float(alpha), float(beta), torch.from_numpy(C0).cuda()).cpu().numpy()
    assert got.shape == (M, N) and np.array_equal(np.ascontiguousarray(got).view(np.uint32), np.ascontiguousarray(want).view(np.uint32))
    got2 = spmm(A, torch.from_numpy(B).cuda()).cpu().numpy()          # alpha=1, beta=0, cached engine
    ref2 = np.zeros(M * Np, np.float32)
    oracle.spmm(M, Np, K, np.float32(1), rp, ci, v, np.ascontiguousarray(Bp.T).reshape(-1), np.float32(0), ref2)
    assert np.array_equal(np.ascontiguousarray(got2).view(np.uint32), np.ascontiguousarray(ref2.reshape(Np, M).T[:, :N]).view(np.uint32))


@pytest.mark.parametrize("N", [16, 24, 128])
@pytest.mark.parametrize("numbering", ["grid", "random"])
def test_op_on_rowmajor_tensors_without_copies(sx, oracle, N, numbering):
    """Round 5: the op hands contiguous torch tensors to sextans_spmm_device_rm where they lie (natural + graph-clustered plans),
    writes into `out` / in place, and stays bit-identical to cpu_spmm_CSR."""
    import torch
    from sextans_amd import api, meshgen, torch_op
    rp, ci, v = api.gen_fem3d_host(18, 17, 16, 3, 7)
    M = K = 18 * 17 * 16 * 3
    if numbering == "random":
        rp, ci, v = meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M // 3, 3, 4))
    A = torch.sparse_csr_tensor(torch.from_numpy(rp.astype(np.int64)), torch.from_numpy(ci.astype(np.int64)), torch.from_numpy(v), size=(M, K)).cuda()
    rs = np.random.RandomState(N)
    B = rs.uniform(-1, 1, (K, N)).astype(np.float32); C0 = rs.uniform(-1, 1, (M, N)).astype(np.float32)
    alpha, beta = np.float32(0.85), np.float32(-2.06)
    want = np.ascontiguousarray(C0.T).reshape(-1).copy()
    oracle.spmm(M, N, K, alpha, rp, ci, v, np.ascontiguousarray(B.T).reshape(-1), beta, want)
    want = np.ascontiguousarray(want.reshape(N, M).T)
    torch_op.clear_cache()
    tB, tC = torch.from_numpy(B).cuda(), torch.from_numpy(C0).cuda()
    got = torch_op.spmm(A, tB, float(alpha), float(beta), tC)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)) and np.array_equal(tC.cpu().numpy(), C0)   # C untouched
    eng = next(iter(torch_op._cache.values()))[0]
    assert "rowmajor" in eng.last_kernel(), eng.last_kernel()
    out = torch.empty((M, N), device="cuda")
    assert torch_op.spmm(A, tB, float(alpha), float(beta), tC, out=out) is out
    assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert torch_op.spmm(A, tB, float(alpha), float(beta), tC, out=tC) is tC                                                     # in place
    assert np.array_equal(tC.cpu().numpy().view(np.uint32), want.view(np.uint32))
    torch_op.clear_cache()


def test_engine_cache_follows_the_matrix(sx, oracle):
    """the engine cache must not hand a stale engine to a different matrix that reuses the addresses of
    a dead one, must notice in-place updates of A's values, and must stay bounded."""
    import gc
    import torch
    from sextans_amd import torch_op
    rs = np.random.RandomState(9)
    M, K, N = 300, 200, 8

    def make(seed):
        r = np.random.RandomState(seed)
        rp, ci, v = random_csr(r, M, K, 6, empty_frac=0.0)
        A = torch.sparse_csr_tensor(torch.from_numpy(rp.astype(np.int64)), torch.from_numpy(ci.astype(np.int64)),
                                    torch.from_numpy(v), size=(M, K)).cuda()
        return A, rp, ci, v

    def check(A, rp, ci, v, B):
        want = np.zeros(M * N, np.float32)
        oracle.spmm(M, N, K, np.float32(1), rp, ci, v, np.ascontiguousarray(B.T).reshape(-1), np.float32(0), want)
        got = torch_op.spmm(A, torch.from_numpy(B).cuda()).cpu().numpy()
        assert np.array_equal(np.ascontiguousarray(got).view(np.uint32), np.ascontiguousarray(want.reshape(N, M).T).view(np.uint32))

    B = rs.uniform(-1, 1, (K, N)).astype(np.float32)
    torch_op.clear_cache()
    for seed in range(12):                       # same shapes, matrices created and dropped: results must follow the data
        A, rp, ci, v = make(seed)
        check(A, rp, ci, v, B)
        del A
        gc.collect()
    assert len(torch_op._cache) <= torch_op._MAX_ENGINES
    torch_op.clear_cache()
    assert len(torch_op._cache) == 0
    A, rp, ci, v = make(99)
    check(A, rp, ci, v, B)
    A.values().mul_(2.0)                         # in-place update: version counter changes, fresh engine
    check(A, rp, ci, (v * np.float32(2.0)).astype(np.float32), B)
    keep = [make(100 + i) for i in range(torch_op._MAX_ENGINES + 3)]
    for a in keep:
        check(*a, B)
    assert len(torch_op._cache) <= torch_op._MAX_ENGINES


def test_op_fast_mode_matches_the_fma_chain(sx, oracle):
    """Round 6: spmm(..., fast=True) = SEXTANS_MODE_FAST.  On a matrix without hub rows nothing is re-associated, so the result is the
    oracle's fmaf chain bit for bit; the strict engine of the same matrix stays cached beside it and stays bit-identical to cpu_spmm_CSR."""
    import torch
    from sextans_amd import api, torch_op
    rp, ci, v = api.gen_fem3d_host(14, 13, 12, 3, 7)
    M = K = 14 * 13 * 12 * 3
    N = 32
    A = torch.sparse_csr_tensor(torch.from_numpy(rp.astype(np.int64)), torch.from_numpy(ci.astype(np.int64)), torch.from_numpy(v), size=(M, K)).cuda()
    rs = np.random.RandomState(2)
    B = rs.uniform(-1, 1, (K, N)).astype(np.float32); C0 = rs.uniform(-1, 1, (M, N)).astype(np.float32)
    alpha, beta = np.float32(0.85), np.float32(-2.06)
    Bc, Cc = np.ascontiguousarray(B.T).reshape(-1), np.ascontiguousarray(C0.T).reshape(-1)
    strict = Cc.copy(); oracle.spmm(M, N, K, alpha, rp, ci, v, Bc, beta, strict)
    fma = Cc.copy(); oracle.spmm_fma(M, N, K, alpha, rp, ci, v, Bc, beta, fma)
    torch_op.clear_cache()
    tB, tC = torch.from_numpy(B).cuda(), torch.from_numpy(C0).cuda()
    g_fast = torch_op.spmm(A, tB, float(alpha), float(beta), tC, fast=True).cpu().numpy()
    g_strict = torch_op.spmm(A, tB, float(alpha), float(beta), tC).cpu().numpy()
    assert len(torch_op._cache) == 2
    assert np.array_equal(np.ascontiguousarray(g_fast.T).reshape(-1).view(np.uint32), fma.view(np.uint32))
    assert np.array_equal(np.ascontiguousarray(g_strict.T).reshape(-1).view(np.uint32), strict.view(np.uint32))
    torch_op.clear_cache()
