"""Pins the oracle to the REFERENCE itself: oracle/sextans_oracle.c against the reference's own
read_suitsparse_matrix / CSC_2_CSR / cpu_spmm_CSR (oracle/_ref/libsextans_ref.so, built by
oracle/Makefile from /root/reference/src/sparse_helper.h + mmio.h).  Skipped where neither the
prebuilt _ref nor /root/reference exists."""
import glob
import os

import numpy as np
import pytest

from util import ALPHA, BETA, CASES, NASA, bits_equal, formula_B, formula_C, random_csr

FILES = sorted(glob.glob(os.path.join(CASES, "*.mtx"))) + [NASA]


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p) for p in FILES])
def test_loader_bit_identical(oracle, ref, path, capfd):
    R = ref.load(path)
    capfd.readouterr()   # the reference prints progress lines
    err, M, K, nnz, cp, ri, cv = oracle.read_mtx(path, 1)
    assert err == 0 and (M, K, nnz) == (R["M"], R["K"], R["nnz"])
    assert np.array_equal(cp, R["csc"][0]) and np.array_equal(ri, R["csc"][1]) and bits_equal(cv, R["csc"][2])
    rp, ci, v = oracle.csc_to_csr(M, K, cp, ri, cv)
    assert np.array_equal(rp, R["csr"][0]) and np.array_equal(ci, R["csr"][1]) and bits_equal(v, R["csr"][2])


@pytest.mark.parametrize("N", [8, 16, 40])
@pytest.mark.parametrize("seed", [0, 1])
def test_spmm_bit_identical_random(oracle, ref, N, seed):
    rs = np.random.RandomState(100 + seed)
    M, K = 257 + 31 * seed, 300 + 17 * seed
    rp, ci, v = random_csr(rs, M, K, 9, long_rows=1)
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    alpha, beta = np.float32(rs.uniform(-2, 2)), np.float32(rs.uniform(-2, 2))
    c1, c2 = C0.copy(), C0.copy()
    oracle.spmm(M, N, K, alpha, rp, ci, v, B, beta, c1)
    ref.spmm(M, N, K, alpha, rp, ci, v, B, beta, c2)
    assert np.array_equal(c1.view(np.uint32), c2.view(np.uint32))


def test_spmm_bit_identical_nasa(oracle, ref, capfd):
    R = ref.load(NASA)
    capfd.readouterr()
    M, K = R["M"], R["K"]
    rp, ci, v = R["csr"]
    for N in (16, 128):
        for B, C0 in ((oracle.init_B(K, N), oracle.init_C(M, N)), (formula_B(K, N), formula_C(M, N))):
            c1, c2 = C0.copy(), C0.copy()
            oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, c1)
            ref.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, c2)
            assert np.array_equal(c1.view(np.uint32), c2.view(np.uint32))


def test_rows_variant_equals_full(oracle):
    rs = np.random.RandomState(7)
    M, K, N = 100, 80, 16
    rp, ci, v = random_csr(rs, M, K, 6)
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    full = C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, full)
    part = C0.copy()
    oracle.spmm_rows(0, 37, M, N, K, ALPHA, rp, ci, v, B, BETA, part)
    oracle.spmm_rows(37, M, M, N, K, ALPHA, rp, ci, v, B, BETA, part)
    assert np.array_equal(full.view(np.uint32), part.view(np.uint32))
