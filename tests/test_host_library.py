"""Product host library (sextans_amd/csrc/host_mtx.cpp behind include/sextans_amd.h) against the
golden vectors and the oracle -- CPU only, no compute calls on a device."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from util import ALPHA, BETA, CASES, NASA, ROOT, bits_equal, default_C, formula_B, formula_C

MANIFEST = json.load(open(os.path.join(CASES, "manifest.json")))


def test_library_exports_every_declared_symbol(sx):
    """Every function include/sextans_amd.h declares is exported by libsextans_amd.so."""
    hdr = open(os.path.join(ROOT, "include", "sextans_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sextans_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    out = subprocess.run(["nm", "-D", "--defined-only", sx.api.LIB_PATH], capture_output=True, text=True, check=True)
    exported = {ln.split()[-1] for ln in out.stdout.splitlines() if ln.strip()}
    missing = sorted(declared - exported)
    assert not missing, f"declared but not exported: {missing}"
    L = sx.api.lib()
    for name in declared:
        getattr(L, name)


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_loader_matches_reference_golden(sx, name):
    g = np.load(os.path.join(CASES, name + ".npz"))
    path = os.path.join(CASES, name + ".mtx")
    rp, ci, v, M, K, nnz = sx.read_suitsparse_matrix(path, sx.api.FMT_CSR)
    assert (M, K, nnz) == (int(g["M"]), int(g["K"]), int(g["nnz"]))
    assert np.array_equal(rp, g["csr_ptr"]) and np.array_equal(ci, g["csr_idx"]) and bits_equal(v, g["csr_val"])
    cp, ri, cv, _, _, _ = sx.read_suitsparse_matrix(path, sx.api.FMT_CSC)
    assert np.array_equal(cp, g["csc_ptr"]) and np.array_equal(ri, g["csc_idx"]) and bits_equal(cv, g["csc_val"])
    rp2, ci2, v2 = sx.CSC_2_CSR(M, K, nnz, cp, ri, cv)
    assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and bits_equal(v2, v)


def test_loader_nasa4704(sx, oracle):
    rp, ci, v, M, K, nnz = sx.read_suitsparse_matrix(NASA)
    M2, K2, nnz2, rp2, ci2, v2 = oracle.load_csr(NASA)
    assert (M, K, nnz) == (M2, K2, nnz2) == (4704, 4704, 104756)
    assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2) and bits_equal(v, v2)


def test_loader_errors_are_codes_not_exit(sx, tmp_path):
    def w(name, text):
        p = tmp_path / name
        p.write_text(text)
        return str(p)
    E = sx.SextansError
    hdr = "%%MatrixMarket matrix coordinate real general\n"
    cases = [
        (str(tmp_path / "missing.mtx"), 1),
        (w("a.mtx", "%%NotMM matrix coordinate real general\n1 1 1\n1 1 1.0\n"), 2),
        (w("a2.mtx", "%%MatrixMarket matrix coordinate real\n1 1 1\n1 1 1.0\n"), 2),
        (w("a3.mtx", "%%MatrixMarket matrix coordinate quaternion general\n1 1 1\n1 1 1.0\n"), 2),
        (w("b.mtx", "%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n"), None),
        (w("c.mtx", "%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 1 0\n"), 5),
        (w("d.mtx", hdr + "2 2 1\n0 1 1.0\n"), 6),
        (w("d2.mtx", hdr + "2 2 1\n3 1 1.0\n"), 6),      # ours only: the reference has no upper check
        (w("e.mtx", hdr + "2 2 2\n1 1 1.0\n"), 8),       # truncated
        (w("f.mtx", hdr + "2 2 1\n1 x 1.0\n"), 8),
        (w("g.mtx", hdr), 3),
    ]
    for path, code in cases:
        with pytest.raises(E) as ei:
            sx.read_suitsparse_matrix(path)
        if code is not None:
            assert ei.value.code == code, (path, ei.value.code)
        else:
            assert ei.value.code in (3, 4)
    # zero-valued entries are dropped BEFORE the index check (sparse_helper.h:145-149)
    rp, ci, v, M, K, nnz = sx.read_suitsparse_matrix(w("h.mtx", hdr + "2 2 2\n0 0 0.0\n2 2 3.5\n"))
    assert nnz == 1 and list(rp) == [0, 0, 1] and v[0] == 3.5
    # "%%MatrixMarket" is a prefix match (strncmp, mmio.h:279)
    assert sx.read_suitsparse_matrix(w("i.mtx", "%%MatrixMarketXYZ matrix coordinate real general\n1 1 1\n1 1 2\n"))[5] == 1


def test_dense_init_verify_gflops_roundup(sx, oracle):
    for (M, K, N) in [(4704, 4704, 16), (7, 5, 8), (13965, 100, 24)]:
        assert bits_equal(sx.init_dense_B(K, N), oracle.init_B(K, N))
        assert bits_equal(sx.init_dense_C(M, N), oracle.init_C(M, N))
        assert bits_equal(sx.init_dense_C(M, N), default_C(M, N))
    assert [sx.round_up_n(n) for n in (1, 8, 9, 16, 17, 127, 128)] == [8, 8, 16, 16, 24, 128, 128]
    rs = np.random.RandomState(3)
    a = rs.uniform(-1, 1, 600).astype(np.float32)
    b = a + (rs.uniform(-1, 1, 600) * 3e-4 * (rs.rand(600) < 0.3)).astype(np.float32)
    assert sx.verify(100, 6, a, b) == pytest.approx(oracle.verify(100, 6, a, b))
    assert sx.verify(100, 6, a, b)[0] > 0
    assert sx.gflops(4704, 16, 104756, 2e-3) == oracle.gflops(4704, 16, 104756, 2e-3)


def test_selfcheck_golden_matches_oracle(sx, oracle):
    """The CLI's CPU golden (used only for its built-in self check) is bit-identical to the oracle."""
    import ctypes as C
    rs = np.random.RandomState(11)
    from util import random_csr
    M, K, N = 300, 280, 24
    rp, ci, v = random_csr(rs, M, K, 8, long_rows=1)
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    c1, c2 = C0.copy(), C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, c1)
    assert sx.api.lib().sextans_selfcheck_golden(M, N, K, ALPHA, rp, ci, v, B, BETA, c2) == 0
    assert np.array_equal(c1.view(np.uint32), c2.view(np.uint32))


def test_no_device_fails_loudly(sx):
    """Without a gfx950 device the compute entry points refuse: there is no CPU fallback."""
    if sx.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(sx.SextansError) as ei:
        sx.Engine(0)
    assert ei.value.code == sx.api.ERR_NO_DEVICE
    c = np.zeros(8, np.float32)
    with pytest.raises(sx.SextansError) as ei:
        sx.spmm_csr(1, 8, 1, 1, 1.0, np.array([0, 1], np.int32), np.array([0], np.int32),
                    np.array([1.0], np.float32), np.ones(8, np.float32), 0.0, c)
    assert ei.value.code == sx.api.ERR_NO_DEVICE


def test_cli_usage_and_errors(sx):
    cli = sx.api.CLI_PATH
    r = subprocess.run([cli], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage:" in r.stdout and "[matrix A file] [N] [rp_time] [alpha] [beta]" in r.stdout
    r = subprocess.run([cli, "a", "b", "c", "d", "e", "f"], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage:" in r.stdout
    r = subprocess.run([cli, "/nonexistent.mtx", "16"], capture_output=True, text=True)
    assert r.returncode == 1 and "Could not open /nonexistent.mtx" in r.stdout
    assert "N = 16" in r.stdout and "alpha = 0.85" in r.stdout and "beta = -2.06" in r.stdout
    r = subprocess.run([cli, "/nonexistent.mtx", "13", "3", "1.5", "-0.25"], capture_output=True, text=True)
    assert "N = 16" in r.stdout and "alpha = 1.5" in r.stdout and "beta = -0.25" in r.stdout


def test_symmetric_file_with_rectangular_size_line(sx, tmp_path):
    """a `symmetric` banner over an M != K size line.  The reference mirrors (c, r) into arrays sized for
    (r, c) (sparse_helper.h:155-161, undefined behaviour); here the mirrored entry is range-checked like any other."""
    from sextans_amd import api
    bad = tmp_path / "sym_rect_bad.mtx"
    bad.write_text("%%MatrixMarket matrix coordinate real symmetric\n2 5 2\n1 1 3.0\n1 5 1.0\n")
    with pytest.raises(api.SextansError) as e:
        api.read_suitsparse_matrix(str(bad))
    assert e.value.code == 6                                       # SEXTANS_ERR_INDEX: (5, 1) does not exist in a 2 x 5 matrix
    with pytest.raises(api.SextansError):
        api.read_suitsparse_matrix(str(bad), fmt=api.FMT_CSC)
    ok = tmp_path / "sym_rect_ok.mtx"                              # every mirrored entry fits: loads, mirrored
    ok.write_text("%%MatrixMarket matrix coordinate real symmetric\n2 5 2\n1 1 3.0\n2 1 4.0\n")
    rp, ci, v, M, K, nnz = api.read_suitsparse_matrix(str(ok))
    assert (M, K, nnz) == (2, 5, 3) and list(rp) == [0, 2, 3] and list(ci) == [0, 1, 0] and list(v) == [3.0, 4.0, 4.0]


def test_round5_entry_points_reject_bad_arguments(sx):
    """The entry points added in round 5 validate before they touch a device (error codes, never exit(); no GPU needed)."""
    import ctypes as C
    from sextans_amd import api
    L = api.lib()
    L.sextans_spmm_device_rm.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    L.sextans_export_row_order.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.sextans_spmm_device_rm(None, 16, 1.0, None, 16, 0.0, None, 16, None, 16, None) == 9          # SEXTANS_ERR_INVALID
    assert L.sextans_export_row_order(None, None, None) == 9
    rr = np.array([0, 64], np.int32)
    assert L.sextans_dist_spmm_rm(None, None, 1, 0, rr, 16, 1.0, None, 16, 0.0, None, 16, None, 16, None) == 9
    assert L.sextans_dist_spmm_bell(None, None, 1, 0, rr, 32, 1.0, None, 64, 0.0, None, 64, None, 64, None) == 9
    assert L.sextans_spmm_bell_device2(None, 32, 1.0, None, 64, 0.0, None, 64, None, 64, None) == 9
    L.sextans_mtx_write.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.sextans_mtx_write(None, 1, 1, None, None, None) == 9
    assert L.sextans_mtx_write(b"/nonexistent-directory/x.mtx", 0, 0, (C.c_int * 1)(0), None, None) == 1  # SEXTANS_ERR_OPEN
    prp = np.array([0, 1, 2], np.int32); pci = np.array([0, 5], np.int32)                                   # column 5 outside a 2-column pattern
    with pytest.raises(api.SextansError):
        api.gen_kron_host(3, prp, pci, 2, 0, 1)
    with pytest.raises(api.SextansError):
        api.gen_kron_host(3, prp, np.array([0, 1], np.int32), 2, 7, 1)                                      # unknown variant


def test_collectives_library_binding_without_a_gpu(sx):
    """sextans_dist_bind_library (round 6) needs no device: an unloadable path is SEXTANS_ERR_STATE (never a silent fallback to another
    library), the loopback communicator of the multi-rank tests (tests/fake_rccl.cpp, cross-compiled here) exports RCCL's entry points and
    hands out its own ids, and the default search can be restored."""
    import ctypes as C
    from loopback_worker import fake_rccl_path
    api = sx.api
    with pytest.raises(api.SextansError) as ei:
        api.dist_bind_library("/nonexistent/librccl_of_nobody.so")
    assert ei.value.code == 12
    path = fake_rccl_path()
    fake = C.CDLL(path)
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllGather", "ncclBroadcast", "ncclGroupStart", "ncclGroupEnd",
                 "ncclGetErrorString"):
        assert hasattr(fake, name), name
    api.dist_bind_library(path)
    uid = api.dist_unique_id()
    assert bytes(uid)[:8] == b"loopback"
    try:
        api.dist_bind_library(None)          # back to the default search (RCCL is part of the image; a box without it reports ERR_STATE)
    except api.SextansError as ex:
        assert ex.code == 12
