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
