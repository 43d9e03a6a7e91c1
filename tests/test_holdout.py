"""The holdout class kron(T_n, P) of round 5 (sextans_amd/holdout.py, csrc/synth.hip kind 5): host generator against scipy."""
import numpy as np
import pytest


def _scipy_kron(n, prp, pci, pm, pk):
    import scipy.sparse as sp
    P = sp.csr_matrix((np.ones(len(pci), np.float32), pci, prp), shape=(pm, pk))
    T = sp.diags([np.ones(n - 1), np.ones(n), np.ones(n - 1)], [-1, 0, 1], format="csr") if n > 1 else sp.identity(1, format="csr")
    A = sp.kron(T, P, format="csr")
    A.sort_indices()
    return A


def test_kron_host_generator_against_scipy(sx):
    from sextans_amd import api, holdout
    prp, pci, pm, pk = holdout.nasa_pattern()
    assert (pm, pk, prp[-1]) == (4704, 4704, 104756)
    n = 5
    rp, ci, v, M, K = holdout.kron_host(n)
    A = _scipy_kron(n, prp, pci, pm, pk)
    assert (M, K) == A.shape == (n * pm, n * pk)
    assert np.array_equal(rp, A.indptr) and np.array_equal(ci, A.indices)
    assert rp[-1] == (3 * n - 2) * 104756
    assert v.min() >= -1 and v.max() < 1 and abs(v.mean()) < 0.01
    # counter based: any row range reproduces the same rows; values are halved off the block diagonal
    rp2, ci2, v2, _ = api.gen_kron_host(n, prp, pci, pk, 0, holdout.SEED, 4000, 9999)
    assert np.array_equal(ci2, ci[rp[4000]:rp[9999]]) and np.array_equal(v2, v[rp[4000]:rp[9999]])
    rows = np.repeat(np.arange(M), np.diff(rp))
    off = (rows // pm) != (ci // pk)
    assert np.abs(v[off]).max() < 0.5 <= np.abs(v[~off]).max()
    # rectangular variant: every third column gone, the rest renumbered, values of the kept entries unchanged
    rpr, cir, vr, Mr, Kr = holdout.kron_host(n, "rect")
    keep = ci % 3 != 2
    assert (Mr, Kr) == (M, 2 * (K // 3) + min(K % 3, 2)) and len(cir) == keep.sum()
    assert np.array_equal(cir, 2 * (ci[keep] // 3) + ci[keep] % 3) and np.array_equal(vr.view(np.uint32), v[keep].view(np.uint32))
    assert np.array_equal(np.diff(rpr), np.bincount(rows[keep], minlength=M))
    # unsymmetric variant: only strictly lower entries go, about 30 % of them
    rpu, ciu, vu, Mu, Ku = holdout.kron_host(n, "unsym")
    rows_u = np.repeat(np.arange(M), np.diff(rpu))
    lower, lower_u = (ci < rows).sum(), (ciu < rows_u).sum()
    assert (Mu, Ku) == (M, K) and (ciu >= rows_u).sum() == (ci >= rows).sum()
    assert 0.68 < lower_u / lower < 0.72
    inner = np.ones(len(ciu), bool); inner[rpu[:-1][np.diff(rpu) > 0]] = False
    assert np.all(np.diff(ciu)[inner[1:]] > 0)
    # argument checks
    with pytest.raises(api.SextansError):
        api.gen_kron_host(n, prp, pci[::-1].copy(), pk, 0, 1)
    with pytest.raises(api.SextansError):
        api.gen_kron_host(0, prp, pci, pk, 0, 1)


def test_mtx_writer_round_trip(sx, tmp_path):
    from sextans_amd import api, holdout
    rp, ci, v, M, K = holdout.kron_host(2, "rect")
    path = str(tmp_path / "k.mtx")
    holdout.write_mtx(path, rp, ci, v, M, K)
    rp2, ci2, v2, M2, K2, nnz2 = api.read_suitsparse_matrix(path)
    assert (M2, K2, nnz2) == (M, K, len(ci))
    assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and np.array_equal(np.asarray(v2).view(np.uint32), v.view(np.uint32))
