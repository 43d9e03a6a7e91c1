"""The oracle (oracle/sextans_oracle.c) against the committed golden vectors.

The goldens in tests/golden/ were produced by the reference's own host functions
(tests/golden/make_golden.py through oracle/_ref); SURVEY.md 8c lists the nasa4704 known answers.
These tests need neither the GPU nor /root/reference."""
import hashlib
import json
import os

import numpy as np
import pytest

from util import ALPHA, BETA, CASES, GOLDEN, NASA, bits_equal, default_C, formula_B, formula_C

MANIFEST = json.load(open(os.path.join(CASES, "manifest.json")))


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_loader_matches_reference_golden(oracle, name):
    g = np.load(os.path.join(CASES, name + ".npz"))
    err, M, K, nnz, cp, ri, cv = oracle.read_mtx(os.path.join(CASES, name + ".mtx"), 1)
    assert err == 0
    assert (M, K, nnz) == (int(g["M"]), int(g["K"]), int(g["nnz"]))
    assert np.array_equal(cp, g["csc_ptr"]) and np.array_equal(ri, g["csc_idx"])
    assert bits_equal(cv, g["csc_val"])
    rp, ci, v = oracle.csc_to_csr(M, K, cp, ri, cv)
    assert np.array_equal(rp, g["csr_ptr"]) and np.array_equal(ci, g["csr_idx"])
    assert bits_equal(v, g["csr_val"])
    # direct CSR read (read_suitsparse_matrix with mf = CSR) gives the same arrays
    err, _, _, _, rp2, ci2, v2 = oracle.read_mtx(os.path.join(CASES, name + ".mtx"), 0)
    assert err == 0 and np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and bits_equal(v2, v)


@pytest.mark.parametrize("name", sorted(MANIFEST))
@pytest.mark.parametrize("N", [8, 24])
def test_spmm_matches_reference_golden(oracle, name, N):
    g = np.load(os.path.join(CASES, name + ".npz"))
    M, K = int(g["M"]), int(g["K"])
    Cm = formula_C(M, N)
    with np.errstate(all="ignore"):
        oracle.spmm(M, N, K, ALPHA, g["csr_ptr"], g["csr_idx"], g["csr_val"], formula_B(K, N), BETA, Cm)
    assert bits_equal(Cm, g[f"C_N{N}"])


def test_nasa4704_known_answers(oracle):
    """nnz = 104756, RowPtr[1..2] = 6, 15, ColIndex[0..4], golden C hashes."""
    known = json.load(open(os.path.join(GOLDEN, "nasa4704_known.json")))
    assert hashlib.sha256(open(NASA, "rb").read()).hexdigest() == known["mtx_sha256"] == \
        "20ad7ce634e138660a367adf7a61541ae9d475359a97d94cac4410374812e784"
    M, K, nnz, rp, ci, v = oracle.load_csr(NASA)
    assert (M, K, nnz) == (4704, 4704, 104756)
    assert [int(rp[1]), int(rp[2])] == [6, 15] and [int(x) for x in ci[:5]] == [0, 1, 23, 24, 33]
    assert int(np.diff(rp).max()) == 42 and np.all(v == 1.0)
    assert hashlib.sha256(rp.tobytes() + ci.tobytes() + v.tobytes()).hexdigest() == known["csr_sha256"]
    want = {16: "988205f823683783aea5cd8c7eb0f88846b59bd957d3429ba63e2110bd3ad88f",
            128: "0a6b46a1581cfd04de887ebb3c815eebf7c93b9dca17e0010bb1fb869f51817e"}
    for N in (16, 128):
        B = oracle.init_B(K, N)
        Cm = oracle.init_C(M, N)
        assert np.all(B == 1.0) and bits_equal(Cm, default_C(M, N))
        oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, Cm)
        assert hashlib.sha256(Cm.tobytes()).hexdigest() == want[N] == known[f"N{N}"]["sha256"]
        assert abs(float(Cm.astype(np.float64).sum()) - known[f"N{N}"]["sum"]) < 1e-6
        for key, val in known[f"N{N}"]["samples"].items():
            m, n = (int(t) for t in key.split(","))
            assert float(Cm[m + M * n]) == val
    g = np.load(os.path.join(GOLDEN, "nasa4704_N16.npz"))
    Cf = formula_C(M, 16)
    oracle.spmm(M, 16, K, ALPHA, rp, ci, v, formula_B(K, 16), BETA, Cf)
    assert bits_equal(Cf, g["C_formula"])


def test_verify_and_gflops(oracle):
    a = np.array([1.0, 2.0, 0.0, -3.0, 1e-5, 100.0], np.float32)
    b = np.array([1.0, 2.0003, 5e-9, -3.0, 2e-5, 100.02], np.float32)
    # |d|/(min+1e-4) > 1e-4: elements 1, 4, 5 mismatch; element 2: 5e-9/1e-4 = 5e-5 passes (the 1e-4 floor)
    n, pct = oracle.verify(3, 2, a, b)
    assert n == 3 and abs(pct - 50.0) < 1e-4
    # sextans-host.cpp:255-260
    assert oracle.gflops(4704, 16, 104756, 1e-3) == pytest.approx(2.0 * 16 * (104756 + 4704) / 1e9 / 1e-3)


def test_loader_error_codes(oracle, tmp_path):
    def w(name, text):
        p = tmp_path / name
        p.write_text(text)
        return str(p)
    assert oracle.read_mtx(str(tmp_path / "missing.mtx"))[0] == 1
    assert oracle.read_mtx(w("a.mtx", "%%NotMM matrix coordinate real general\n1 1 1\n1 1 1.0\n"))[0] == 2
    assert oracle.read_mtx(w("b.mtx", "%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n"))[0] in (3, 4)
    assert oracle.read_mtx(w("c.mtx", "%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 1 0\n"))[0] == 5
    assert oracle.read_mtx(w("d.mtx", "%%MatrixMarket matrix coordinate real general\n2 2 1\n0 1 1.0\n"))[0] == 6
