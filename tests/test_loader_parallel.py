"""Loader throughput work: the parallel tokenizer / float parser / bucket sort must give
exactly the arrays of the fscanf-based loader (oracle restatement of sparse_helper.h:112-259; the
reference itself when oracle/_ref is present) for every thread count and line layout."""
import os
from decimal import Decimal, getcontext

import numpy as np
import pytest

import sextans_amd.api as api
from oracle.bindings import CSC, CSR, Ref
from util import bits_equal

HDR = "%%MatrixMarket matrix coordinate real general\n% c\n"


@pytest.fixture
def threads():
    old = os.environ.get("SEXTANS_LOADER_THREADS")

    def set_threads(n):
        os.environ["SEXTANS_LOADER_THREADS"] = str(n)
    yield set_threads
    if old is None:
        os.environ.pop("SEXTANS_LOADER_THREADS", None)
    else:
        os.environ["SEXTANS_LOADER_THREADS"] = old


def same_as_oracle(oracle, path, threads, counts=(1, 2, 5, 16)):
    for fmt, ofmt in ((api.FMT_CSR, CSR), (api.FMT_CSC, CSC)):
        err, M, K, nnz, p, i, v = oracle.read_mtx(path, ofmt)
        assert err == 0
        for t in counts:
            threads(t)
            gp, gi, gv, gM, gK, gnnz = api.read_suitsparse_matrix(path, fmt)
            assert (gM, gK, gnnz) == (M, K, nnz), t
            assert np.array_equal(gp, p) and np.array_equal(gi, i) and bits_equal(gv, v), (t, fmt)


def test_layouts_that_are_not_one_entry_per_line(oracle, tmp_path, threads):
    rs = np.random.RandomState(3)
    n, M, K = 4000, 300, 257
    r, c = rs.randint(1, M + 1, n), rs.randint(1, K + 1, n)
    v = rs.uniform(-2, 2, n).astype(np.float32)
    toks = []
    for a, b, x in zip(r, c, v):
        toks += [str(a), str(b), repr(float(x))]
    seps = [" ", "\n", "\t", "\r\n", "  \n ", "\n\n", " \t "]
    body = "".join(t + seps[rs.randint(len(seps))] for t in toks).rstrip()     # no trailing newline
    p = tmp_path / "tokens.mtx"
    p.write_text(HDR + f"{M} {K} {n}\n" + body)
    same_as_oracle(oracle, str(p), threads)
    # symmetric + duplicates + explicit zeros (+0 dropped, -0 kept), several entries per line
    lines = []
    for a, b, x in zip(r[:1500], c[:1500], v[:1500]):
        a, b = max(a, b), min(a, b)
        b = min(b, 257)
        lines.append(f"{a} {b} {0.0 if x > 1.5 else (-0.0 if x < -1.5 else float(x))!r}")
    lines += lines[:40]
    p2 = tmp_path / "sym.mtx"
    p2.write_text("%%MatrixMarket matrix coordinate real symmetric\n" + f"{M} {M} {len(lines)}\n" +
                  "  ".join(lines) + "\n")
    same_as_oracle(oracle, str(p2), threads)
    # pattern file, more tokens than the header announces (extra ones are ignored)
    p3 = tmp_path / "pat.mtx"
    p3.write_text("%%MatrixMarket matrix coordinate pattern general\n" + f"{M} {K} 1000\n" +
                  "\n".join(f"{a} {b}" for a, b in zip(r, c)) + "\n")
    same_as_oracle(oracle, str(p3), threads)


def test_float_text_is_rounded_once(oracle, tmp_path, threads):
    """strtof semantics: decimal text -> nearest float directly.  Includes values on and next to float
    rounding boundaries (where decimal -> double -> float would round twice), long mantissas, exponents,
    subnormals, hex, inf."""
    getcontext().prec = 80
    rs = np.random.RandomState(7)
    texts = []
    f = rs.uniform(-4, 4, 300).astype(np.float32)
    for x in f:
        lo = Decimal(float(x))
        hi = Decimal(float(np.nextafter(x, np.float32(np.inf))))
        mid = (lo + hi) / 2
        for d in (mid, mid + Decimal("1e-40"), mid - Decimal("1e-40"), mid + Decimal("1e-17"), lo, hi):
            texts.append(format(d, "f"))
            texts.append(format(d, ".30e"))
    texts += ["%.*g" % (rs.randint(1, 20), x) for x in rs.uniform(-1e6, 1e6, 2000)]
    texts += ["%.*e" % (rs.randint(0, 18), x) for x in 10.0 ** rs.uniform(-44, 38, 2000)]
    texts += ["1e-46", "1.4e-45", "1e-39", "3.4028235e38", "3.4028236e38", "1e39", "inf", "-inf", "0x1.8p1",
              "00012.5000", ".5", "5.", "+7", "-0.0", "1E5", "1e+05", "123456789012345678901234567890",
              "0.000000000000000000000000000000000000000000001", "16777217", "9007199254740993"]
    n = len(texts)
    p = tmp_path / "floats.mtx"
    p.write_text(HDR + f"{n} 3 {n}\n" + "".join(f"{i + 1} {i % 3 + 1} {t}\n" for i, t in enumerate(texts)))
    same_as_oracle(oracle, str(p), threads, counts=(1, 4))


@pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_against_the_reference_loader_itself(tmp_path, threads):
    rs = np.random.RandomState(11)
    n, M, K = 30000, 5000, 4000
    p = tmp_path / "ref.mtx"
    with open(p, "w") as fh:
        fh.write("%%MatrixMarket matrix coordinate real symmetric\n" + f"{M} {M} {n}\n")
        for a, b, x in zip(rs.randint(1, M + 1, n), rs.randint(1, M + 1, n), rs.uniform(-1, 1, n)):
            fh.write(f"{max(a, b)} {min(a, b)} {x:.9g}\n")
    d = Ref().load(str(p))
    for t in (1, 3, 8):
        threads(t)
        gp, gi, gv, gM, gK, gnnz = api.read_suitsparse_matrix(str(p), api.FMT_CSR)
        assert np.array_equal(gp, d["csr"][0]) and np.array_equal(gi, d["csr"][1]) and bits_equal(gv, d["csr"][2])
        gp, gi, gv, _, _, _ = api.read_suitsparse_matrix(str(p), api.FMT_CSC)
        assert np.array_equal(gp, d["csc"][0]) and np.array_equal(gi, d["csc"][1]) and bits_equal(gv, d["csc"][2])


def test_first_error_in_file_order_wins(tmp_path, threads):
    body = ["1 1 1.0"] * 2000
    body[700] = "9 1 1.0"          # row out of range  -> code 6
    body[1500] = "1 x 1.0"         # malformed         -> code 8
    p = tmp_path / "e.mtx"
    p.write_text(HDR + "4 4 2000\n" + "\n".join(body) + "\n")
    for t in (1, 7):
        threads(t)
        with pytest.raises(api.SextansError) as ei:
            api.read_suitsparse_matrix(str(p))
        assert ei.value.code == 6
    body[300] = "1 1 abc"
    p.write_text(HDR + "4 4 2000\n" + "\n".join(body) + "\n")
    for t in (1, 7):
        threads(t)
        with pytest.raises(api.SextansError) as ei:
            api.read_suitsparse_matrix(str(p))
        assert ei.value.code == 8
    p.write_text(HDR + "4 4 2000\n" + "\n".join(body[:100]) + "\n")            # ends early
    threads(4)
    with pytest.raises(api.SextansError) as ei:
        api.read_suitsparse_matrix(str(p))
    assert ei.value.code == 8


def test_binary_container_and_cached_read(tmp_path, oracle):
    src = tmp_path / "m.mtx"
    rs = np.random.RandomState(5)
    n, M, K = 5000, 400, 300
    src.write_text(HDR + f"{M} {K} {n}\n" + "".join(
        f"{a} {b} {x:.8g}\n" for a, b, x in zip(rs.randint(1, M + 1, n), rs.randint(1, K + 1, n),
                                                rs.uniform(-1, 1, n))))
    for fmt, suffix in ((api.FMT_CSR, ".csr.sxbin"), (api.FMT_CSC, ".csc.sxbin")):
        plain = api.read_suitsparse_matrix(str(src), fmt)
        first = api.read_suitsparse_matrix(str(src), fmt, cache=True)
        assert api.read_suitsparse_matrix.last_cache_hit is False
        cpath = str(src) + suffix
        assert os.path.getsize(cpath) == 64 + 4 * len(plain[0]) + 8 * plain[5]
        second = api.read_suitsparse_matrix(str(src), fmt, cache=True)
        assert api.read_suitsparse_matrix.last_cache_hit is True
        for a, b, c in zip(plain, first, second):
            assert np.array_equal(a, b) and np.array_equal(a, c)
        got = api.matrix_load(cpath)
        assert got[0] == fmt and np.array_equal(got[1], plain[0]) and bits_equal(got[3], plain[2])
    # a changed source invalidates the container (size / mtime recorded in its header)
    src.write_text(HDR + "2 2 1\n1 2 3.5\n")
    rp, ci, v, M2, K2, nnz2 = api.read_suitsparse_matrix(str(src), api.FMT_CSR, cache=True)
    assert api.read_suitsparse_matrix.last_cache_hit is False
    assert (M2, K2, nnz2) == (2, 2, 1) and list(rp) == [0, 1, 1] and list(ci) == [1] and list(v) == [3.5]
    # a container of the other format under an explicit path is not used, it is rewritten
    explicit = str(tmp_path / "x.bin")
    api.read_suitsparse_matrix(str(src), api.FMT_CSC, cache=explicit)
    api.read_suitsparse_matrix(str(src), api.FMT_CSR, cache=explicit)
    assert api.read_suitsparse_matrix.last_cache_hit is False
    assert api.matrix_load(explicit)[0] == api.FMT_CSR
    # damaged containers are rejected and rebuilt
    with open(explicit, "r+b") as fh:
        fh.truncate(70)
    with pytest.raises(api.SextansError) as ei:
        api.matrix_load(explicit)
    assert ei.value.code == 8
    api.read_suitsparse_matrix(str(src), api.FMT_CSR, cache=explicit)
    assert api.matrix_load(explicit)[6] == 1
    # stand-alone save / load (no source file behind it)
    api.matrix_save(str(tmp_path / "s.bin"), api.FMT_CSR, 3, 4, np.array([0, 1, 1, 2], np.int32),
                    np.array([3, 0], np.int32), np.array([1.5, -2], np.float32))
    f, p, i, v, M3, K3, nnz3 = api.matrix_load(str(tmp_path / "s.bin"))
    assert (f, M3, K3, nnz3) == (api.FMT_CSR, 3, 4, 2) and list(p) == [0, 1, 1, 2] and list(i) == [3, 0]
    with pytest.raises(api.SextansError):
        api.read_suitsparse_matrix(str(tmp_path / "missing.mtx"), api.FMT_CSR, cache=True)
