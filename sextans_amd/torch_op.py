"""Operator-style embedding: torch.sparse_csr in, dense out, over the C ABI.

    C = spmm(A_csr, B, alpha=1.0, beta=0.0, C=None)

A: torch.sparse_csr_tensor (fp32 values, int32/int64 indices) on a GPU; B: dense (K, N) fp32; returns a
dense (M, N) tensor.  Torch tensors are ROW-major, and so is the entry point used here (sextans_spmm_device_rm, round 5): a
contiguous fp32 B with N % 8 == 0 is handed to the kernel where it lies -- for N = 16 it IS the kernel's B panel -- and the
result is written straight into the (M, N) tensor that is returned: no transposes, no copies.  (Until round 4 this op transposed B
into a column-major copy that the engine repacked into row-major panels again, and C likewise in reverse: two passes over B and two
over C per call.)  N that is not a multiple of 8 is padded up internally (the reference's N-tile granularity, sextans-host.cpp:51).  One cached engine per live
matrix (keyed on tensor addresses + version counters; the entry pins the tensors, at most 8 kept, clear_cache()
drops them); not part of the reference, whose only front end is the CLI.
"""
import collections

import torch

from . import api

_MAX_ENGINES = 8
_cache = collections.OrderedDict()     # key -> (engine, arrays the engine reads, A's own index/value tensors)


def _evict(key):
    ent = _cache.pop(key, None)
    if ent is not None:
        ent[0].close()


def clear_cache():
    """Drop every cached engine (and the references that pin the matrices they were built from)."""
    for key in list(_cache):
        _evict(key)


def _engine_for(A, dev, fast=False):
    """One engine per sparse matrix, at most _MAX_ENGINES of them (LRU: every entry pins an engine with its device
    workspaces).  The key holds the addresses AND the version counters of A's three tensors, so an in-place update
    of A gets a fresh engine (the packed forms snapshot the values); tensors without version counters (inference mode)
    get a fresh engine every time.  The entry also keeps A's own tensors alive:
    as long as it exists their storage cannot be freed and handed to a different matrix, so equal addresses always
    mean the same matrix."""
    crow, col, val = A.crow_indices(), A.col_indices(), A.values()
    M, K = A.shape
    def ver(t):   # inference-mode tensors have no version counter
        try:
            return t._version
        except RuntimeError:
            return None
    vers = (ver(crow), ver(col), ver(val))
    key = (dev, crow.data_ptr(), col.data_ptr(), val.data_ptr(), vers, M, K, val.numel(), bool(fast))
    # Without version counters an in-place update of A (same storage, same addresses) cannot be told from no update, and the
    # packed forms snapshot the values: such matrices are never served from the cache -- a stale entry under the same
    # addresses is dropped and the engine is rebuilt on every call.
    cacheable = None not in vers
    ent = _cache.get(key)
    if ent is not None:
        if cacheable:
            _cache.move_to_end(key)
            return ent[0]
        _evict(key)
    crow32, col32 = crow.to(torch.int32).contiguous(), col.to(torch.int32).contiguous()
    val32 = val.to(torch.float32).contiguous()
    eng = api.Engine(dev)
    if fast:   # SEXTANS_MODE_FAST: FMA + re-associated hub rows, |d| <= 1e-4 * (|alpha| sum|a b| + |beta c|); the default is bit identity with cpu_spmm_CSR
        eng.set_option("mode", 1)
    eng.set_matrix_csr_device(M, K, val32.numel(), crow32.data_ptr(), col32.data_ptr(), val32.data_ptr())
    _cache[key] = (eng, (crow32, col32, val32), (crow, col, val))   # the engine does not copy: keep its arrays alive
    while len(_cache) > _MAX_ENGINES:
        _evict(next(iter(_cache)))
    return eng


def _rowmajor(t, rows, cols, colsp):
    """fp32, unit column stride, 16-byte aligned base and row stride: the tensor itself when it qualifies, else a padded copy"""
    if (t.dtype == torch.float32 and cols == colsp and t.stride(1) == 1 and t.stride(0) >= cols and t.stride(0) % 4 == 0 and
            t.data_ptr() % 16 == 0):
        return t
    out = torch.zeros((rows, colsp), dtype=torch.float32, device=t.device)
    out[:, :cols] = t
    return out


def spmm(A, B, alpha=1.0, beta=0.0, C=None, out=None, fast=False):
    """out (optional): an (M, N) fp32 row-major tensor that receives the result (N % 8 == 0); may be C itself (in place).
    fast (round 6): the engine's documented in-tolerance mode (include/sextans_amd.h, SEXTANS_MODE_FAST) instead of bit identity with the
    reference's cpu_spmm_CSR; a matrix used in both modes keeps one engine per mode."""
    if A.layout != torch.sparse_csr or not A.is_cuda or not B.is_cuda:
        raise TypeError("spmm expects a CUDA/HIP torch.sparse_csr matrix and a CUDA/HIP dense B")
    M, K = A.shape
    if B.dim() != 2 or B.shape[0] != K:
        raise ValueError("shape mismatch")
    N = B.shape[1]
    Np = api.round_up_n(N)
    dev = A.device.index or 0
    eng = _engine_for(A, dev, fast)
    Brm = _rowmajor(B, K, N, Np)
    if C is not None and beta != 0.0:
        if tuple(C.shape) != (M, N):
            raise ValueError("shape mismatch")
        Cin = _rowmajor(C, M, N, Np)
    else:
        Cin = None
    if out is not None and (tuple(out.shape) != (M, N) or _rowmajor(out, M, N, Np) is not out):
        raise ValueError("out must be an (M, N) fp32 row-major tensor with N % 8 == 0, 16-byte aligned")
    Cout = out if out is not None else (Cin if (Cin is not None and Cin is not C) else None)
    if Cout is None:       # beta * C_in with C_in = 0 when no C is given: zeros, also for beta == 0 (0 * NaN would not be 0)
        Cout = torch.zeros((M, Np), dtype=torch.float32, device=B.device) if Cin is None else torch.empty((M, Np), dtype=torch.float32, device=B.device)
    if Cin is None:
        if out is not None:
            out.zero_()
        Cin = Cout
    stream = torch.cuda.current_stream(B.device).cuda_stream
    eng.spmm_device_rm(Np, float(alpha), Brm.data_ptr(), Brm.stride(0), float(beta), Cin.data_ptr(), Cin.stride(0), Cout.data_ptr(),
                       Cout.stride(0), stream)
    return Cout if Np == N else Cout[:, :N]
