"""Shared test helpers: deterministic dense operands and comparison utilities."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = os.path.join(GOLDEN, "cases")
NASA = os.path.join(ROOT, "matrices", "nasa4704", "nasa4704.mtx")

ALPHA = np.float32(0.85)    # sextans-host.cpp:29
BETA = np.float32(-2.06)    # sextans-host.cpp:30


def formula_B(K, N):
    """Column-major K x N, asymmetric in (k, n), multiples of 1/16 in [-1, 1): catches row/column
    swaps and B-indexing errors that the reference's B == 1.0 self-check cannot see."""
    k = np.arange(K, dtype=np.int64)[:, None]
    n = np.arange(N, dtype=np.int64)[None, :]
    B = ((k * 7 + n * 13 + (k * n) % 5) % 31 - 15).astype(np.float32) / np.float32(16)
    return np.ascontiguousarray(B.T).reshape(-1)      # B[k + K*n]


def formula_C(M, N):
    m = np.arange(M, dtype=np.int64)[:, None]
    n = np.arange(N, dtype=np.int64)[None, :]
    Cm = ((m * 3 + n * 11 + (m * n) % 7) % 29 - 14).astype(np.float32) / np.float32(8)
    return np.ascontiguousarray(Cm.T).reshape(-1)     # C[m + M*n]


def default_C(M, N):
    """sextans-host.cpp:109 evaluated in double, stored as float."""
    m = np.arange(M, dtype=np.float64)[None, :]
    n = np.arange(N, dtype=np.float64)[:, None]
    return (1.0 * (m + 1) * (n + 1) / M / N).astype(np.float32).reshape(-1)


def bits_equal(a, b):
    """Bit-exact equality of two float32 arrays; NaNs compare equal to NaNs regardless of payload
    (x86 and gfx950 generate different default-NaN sign bits)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return False
    return np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb])


def random_csr(rs, M, K, mean_nnz, empty_frac=0.05, long_rows=0):
    """Random CSR with sorted distinct columns, some empty rows and optional long rows."""
    lens = rs.poisson(mean_nnz, M)
    lens[rs.rand(M) < empty_frac] = 0
    for _ in range(long_rows):
        lens[rs.randint(0, M)] = min(K, int(mean_nnz * 40))
    lens = np.minimum(lens, K)
    rp = np.zeros(M + 1, np.int32)
    rp[1:] = np.cumsum(lens)
    ci = np.empty(rp[-1], np.int32)
    for i in range(M):
        if lens[i]:
            ci[rp[i]:rp[i + 1]] = np.sort(rs.choice(K, size=lens[i], replace=False))
    val = rs.uniform(-1, 1, rp[-1]).astype(np.float32)
    return rp, ci, val
