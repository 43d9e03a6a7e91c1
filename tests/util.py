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


# ---- the accelerator's buffer formats, restated in numpy for the tests ----

def bitrev3(x):
    return ((x & 1) << 2) | (x & 2) | ((x >> 2) & 1)


def edge_words(ptr, row, col, val):
    """edge_list_64bit for 8 channels, restated from sparse_helper.h:406-473: scheduler output
    (row/col/val[64, L], row == -1 = bubble) -> uint64 channels[8, round_up(8 L, 512)].
    word = (col & 0x3FFF) << 50 | (row & 0x3FFFF) << 32 | fp32 bits; bubble = 0x3FFFF << 32;
    PE p -> channel p % 8, slot bitrev3(p / 8) of each 8-word group (:458-464)."""
    L = int(ptr[-1])
    ch = np.zeros((8, (8 * L + 511) // 512 * 512), np.uint64)
    bits = np.ascontiguousarray(val, np.float32).view(np.uint32).astype(np.uint64)
    w = ((col.astype(np.int64).astype(np.uint64) & np.uint64(0x3FFF)) << np.uint64(50)) | \
        ((row.astype(np.int64).astype(np.uint64) & np.uint64(0x3FFFF)) << np.uint64(32)) | bits
    w = np.where(row == -1, np.uint64(0x3FFFF) << np.uint64(32), w)
    for p in range(64):
        ch[p % 8, bitrev3(p // 8):8 * L:8] = w[p]
    return ch


def chan_b_ref(K, N, B, num_ch_b):
    """mat_B_fpga_vec of sextans-host.cpp:152-177, element by element."""
    cs = (K + 15) // 16 * 16 if num_ch_b == 8 else (K + 7) // 8 * 8 * 2
    ln = (cs * (N // 8) + 1023) // 1024 * 1024
    ch = np.zeros((num_ch_b, ln), np.float32)
    for nn in range(N):
        for kk in range(K):
            if num_ch_b == 4:
                ch[(nn // 2) % 4, (kk // 8) * 16 + (nn % 2) * 8 + kk % 8 + cs * (nn // 8)] = B[kk + K * nn]
            else:
                ch[nn % 8, kk + cs * (nn // 8)] = B[kk + K * nn]
    return ch


def chan_c_ref(M, N, Cm):
    """mat_C_fpga_in of sextans-host.cpp:179-195 (same indexing reads mat_C_fpga_vec, :264-270)."""
    cs = (M + 15) // 16 * 16
    ln = (cs * (N // 8) + 1023) // 1024 * 1024
    ch = np.zeros((8, ln), np.float32)
    m = np.arange(M)
    for nn in range(N):
        ch[m % 8, cs * (nn // 8) + (m // 8) * 8 + nn % 8] = Cm[m + M * nn]
    return ch
