"""Holdout inputs with real-matrix structure, at scale (round 5, VERDICT r04 task 1).

Every >= 0.5 roofline number of rounds 1-4 was measured on a grid / mesh generator written in the same rounds as the dispatcher's
heuristics.  The only real SuiteSparse matrix in this mount is nasa4704 (4 704 rows); the reference's evaluation is SuiteSparse
(README.md:17-18,31).  kron(T_n, nasa4704) carries nasa4704's real local structure to 4 M rows: n copies of its pattern on the block
diagonal, each coupled to its neighbours through the same pattern (csrc/synth.hip, kind 5; same bits on host and device).

  numberings   "natural" (as generated), "random" (seeded row/column permutation), "rcm" (reverse Cuthill-McKee, scipy, on the host)
  variants     "" square symmetric pattern, "rect" every third column dropped (M x 2K/3), "unsym" 30 % of the strictly lower entries dropped

Measurement / test infrastructure: nothing on the product path imports this module.
"""
import os

import numpy as np

from . import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NASA = os.path.join(ROOT, "matrices", "nasa4704", "nasa4704.mtx")
SEED = 11
VARIANTS = {"": 0, "sym": 0, "rect": 1, "unsym": 2, "rectunsym": 3}


def nasa_pattern():
    rp, ci, _, M, K, _ = api.read_suitsparse_matrix(NASA)
    return np.asarray(rp, np.int32), np.asarray(ci, np.int32), M, K


def kron_host(n, variant="", seed=SEED, r0=0, r1=None, pattern=None):
    """(row_ptr, col_idx, val, M, K) of kron(T_n, P) on the host."""
    prp, pci, pm, pk = pattern or nasa_pattern()
    rp, ci, v, K = api.gen_kron_host(n, prp, pci, pk, VARIANTS[variant], seed, r0, r1)
    return rp, ci, v, n * pm, K


def rcm_permutation(n, pattern=None):
    """new_of_old of reverse Cuthill-McKee over the SQUARE symmetric kron pattern (scipy; ~1 min at n = 850)."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    prp, pci, pm, pk = pattern or nasa_pattern()
    rp, ci, _, M, K = kron_host(n, "", pattern=(prp, pci, pm, pk))
    G = sp.csr_matrix((np.ones(len(ci), np.int8), ci, rp), shape=(M, K))
    order = reverse_cuthill_mckee(G, symmetric_mode=True)
    new_of_old = np.empty(M, np.int64)
    new_of_old[order] = np.arange(M)
    return new_of_old


def random_permutation(M, seed=1):
    return np.random.RandomState(seed).permutation(M).astype(np.int64)


def kron_device(device, n, variant="", numbering="natural", seed=SEED, pattern=None):
    """(M, K, d_rp, d_ci, d_v, nnz): the matrix in HBM.  Numberings other than "natural" need the square variants ("" / "unsym":
    rows and columns are renumbered together, P A P^T)."""
    prp, pci, pm, pk = pattern or nasa_pattern()
    p, i, v, nnz, K = api.gen_kron_device(device, n, prp, pci, pk, VARIANTS[variant], seed)
    M = n * pm
    if numbering == "natural":
        return M, K, p, i, v, nnz
    if M != K:
        raise ValueError("renumbering needs a square variant")
    perm = random_permutation(M) if numbering == "random" else rcm_permutation(n, (prp, pci, pm, pk)) if numbering == "rcm" else None
    if perm is None:
        raise ValueError("numbering: natural | random | rcm")
    q = api.permute_symmetric_device(device, M, nnz, p, i, v, perm)
    for old in (p, i, v):
        api.device_free(device, old)
    return (M, K) + q + (nnz,)


def write_mtx(path, rp, ci, v, M, K):
    """Matrix-Market `real general` file of a host CSR matrix (sextans_mtx_write: %.9g keeps every fp32 value exactly)."""
    import ctypes as C
    L = api.lib()
    L.sextans_mtx_write.argtypes = [C.c_char_p, C.c_int, C.c_int, api._i32p, api._i32p, api._f32p]
    api._check(L.sextans_mtx_write(path.encode(), M, K, np.ascontiguousarray(rp, np.int32), np.ascontiguousarray(ci, np.int32),
                                   np.ascontiguousarray(v, np.float32)), "mtx_write")
