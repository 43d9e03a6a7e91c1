"""ctypes mirror of include/sextans_amd.h -- the host-side call surface of the engine.

Function names follow the reference's host library (src/sparse_helper.h, src/sextans-host.cpp) so
tests read like the reference's own harness:

    read_suitsparse_matrix  sparse_helper.h:169      CSC_2_CSR   sparse_helper.h:475
    Engine.spmm             cpu_spmm_CSR argument meaning (sparse_helper.h:262) behind the
                            tapa::invoke(Sextans, ...) boundary (sextans-host.cpp:237-251)

All compute goes through libsextans_amd.so (HIP, gfx950).  There is no Python or CPU fallback: a
missing library raises at import of this module's `lib()`; a missing GPU raises SextansError
(SEXTANS_ERR_NO_DEVICE) at Engine creation.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsextans_amd.so")
CLI_PATH = os.path.join(_HERE, "bin", "sextans")

FMT_CSR, FMT_CSC = 0, 1
ERR_NO_DEVICE = 10

_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_lib = None


class Packed(C.Structure):
    """Mirror of struct sextans_packed (include/sextans_amd.h)."""
    _fields_ = [("M", C.c_int), ("K", C.c_int), ("nnz", C.c_int64), ("lanes_per_row", C.c_int),
                ("nblk", C.c_int), ("blk_row", C.POINTER(C.c_int)), ("dict_ptr", C.POINTER(C.c_int)),
                ("dict", C.POINTER(C.c_int)), ("row_off", C.POINTER(C.c_int)),
                ("idx16", C.POINTER(C.c_uint16)), ("col32", C.POINTER(C.c_int)),
                ("val", C.POINTER(C.c_float)), ("stream_len", C.c_int64), ("max_dict", C.c_int),
                ("nnz_in_panel_blocks", C.c_int64)]


class WindowPacked(C.Structure):
    """Mirror of struct sextans_window_packed (include/sextans_amd.h)."""
    _fields_ = [("M", C.c_int), ("K", C.c_int), ("nnz", C.c_int64), ("rows_per_wave", C.c_int),
                ("window_cols", C.c_int), ("nwaves", C.c_int), ("steps", C.c_int64),
                ("padded_lower_bound", C.c_int64), ("wave_step0", C.POINTER(C.c_int)),
                ("stream", C.POINTER(C.c_uint64))]


class Edges(C.Structure):
    """Mirror of struct sextans_edges (include/sextans_amd.h)."""
    _fields_ = [("M", C.c_int32), ("K", C.c_int32), ("num_windows", C.c_int32), ("num_a_len", C.c_int32),
                ("nnz", C.c_int64), ("ptr_len", C.c_int64), ("chan_len", C.c_int64),
                ("edge_list_ptr", C.POINTER(C.c_int32)), ("channel", C.POINTER(C.c_uint64) * 8)]


class SextansError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        msg = lib().sextans_error_string(code).decode()
        detail = lib().sextans_last_error().decode()
        super().__init__(f"{where}: [{code}] {msg}" + (f" ({detail})" if detail else ""))


# Entry points added after round 4: the only ones a comparison against an OLDER build (tools/nasa_ab.py loads the libraries of earlier
# rounds for same-box A/Bs) may lack.  Any other missing name is a typo or a broken build and fails the load.
_OPTIONAL_SYMBOLS = frozenset((
    "sextans_spmm_device_rm", "sextans_dist_spmm_rm", "sextans_spmm_bell_device2", "sextans_dist_spmm_bell", "sextans_profile_read_post",
    "sextans_gen_kron_host", "sextans_gen_kron_device", "sextans_csr_slice_rows_device", "sextans_csr_permute_symmetric_device",
    "sextans_export_row_order", "sextans_mtx_read_cached", "sextans_matrix_save", "sextans_matrix_load",
    "sextans_prepare", "sextans_dist_prepare", "sextans_dist_bind_library", "sextans_device_alloc", "sextans_device_copy"))


class _Optional:
    """ctypes library proxy for lib(): declaring the prototype of an entry point that an OLDER build lacks is skipped instead of
    failing the whole load -- for the names in _OPTIONAL_SYMBOLS only; anything else raises AttributeError (a misspelled name would
    otherwise lose its prototype silently and ctypes would truncate 64-bit arguments to int)."""
    class _Missing:
        argtypes = restype = None
    def __init__(self, L):
        object.__setattr__(self, "_L", L)
    def __getattr__(self, name):
        try:
            return getattr(self._L, name)
        except AttributeError:
            if name in _OPTIONAL_SYMBOLS:
                return _Optional._Missing()
            raise


def lib():
    """Load libsextans_amd.so once.  torch (if importable) is imported first so that this library
    binds to the same libamdhip64.so.7 instance torch uses (one HIP runtime per process)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP engine with `python -m sextans_amd.build` "
            "(there is no CPU fallback)")
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    raw = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    L = _Optional(raw)      # (prototypes of entry points an older build lacks are skipped)
    pi, pf = C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_float))
    ip = C.POINTER(C.c_int)
    L.sextans_error_string.restype = C.c_char_p
    L.sextans_error_string.argtypes = [C.c_int]
    L.sextans_last_error.restype = C.c_char_p
    L.sextans_mtx_read.argtypes = [C.c_char_p, C.c_int, ip, ip, ip, pi, pi, pf]
    L.sextans_mtx_read_cached.argtypes = [C.c_char_p, C.c_char_p, C.c_int, ip, ip, ip, pi, pi, pf, ip]
    L.sextans_matrix_save.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f32p]
    L.sextans_matrix_load.argtypes = [C.c_char_p, ip, ip, ip, ip, pi, pi, pf]
    L.sextans_host_free.argtypes = [C.c_void_p]
    L.sextans_host_free.restype = None
    L.sextans_csc_to_csr.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f32p, _i32p, _i32p,
                                     _f32p]
    L.sextans_init_dense_B.argtypes = [C.c_int, C.c_int, _f32p]
    L.sextans_init_dense_B.restype = None
    L.sextans_init_dense_C.argtypes = [C.c_int, C.c_int, _f32p]
    L.sextans_init_dense_C.restype = None
    L.sextans_round_up_n.argtypes = [C.c_int]
    L.sextans_verify.argtypes = [C.c_int, C.c_int, _f32p, _f32p, C.POINTER(C.c_float)]
    L.sextans_gflops.restype = C.c_double
    L.sextans_gflops.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_double]
    L.sextans_selfcheck_golden.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, _i32p, _i32p,
                                           _f32p, _f32p, C.c_float, _f32p]
    L.sextans_pack_csr.argtypes = [C.c_int, C.c_int, _i32p, _i32p, _f32p, C.c_int, C.c_int, C.POINTER(Packed)]
    L.sextans_packed_free.argtypes = [C.POINTER(Packed)]
    L.sextans_packed_free.restype = None
    L.sextans_unpack_csr.argtypes = [C.POINTER(Packed), _i32p, _i32p, _f32p]
    L.sextans_window_pack_csr.argtypes = [C.c_int, C.c_int, _i32p, _i32p, _f32p, C.c_int, C.c_int,
                                          C.POINTER(WindowPacked)]
    L.sextans_window_packed_free.argtypes = [C.POINTER(WindowPacked)]
    L.sextans_window_packed_free.restype = None
    L.sextans_get_stat.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]
    L.sextans_align_row.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.sextans_reassociated_rows.argtypes = [C.c_void_p, _i32p, C.c_int, C.POINTER(C.c_int)]
    L.sextans_export_plan.argtypes = [C.c_void_p, C.c_int, C.POINTER(Packed)]
    L.sextans_partition_rows_by_nnz.argtypes = [C.c_int, _i32p, C.c_int, _i32p]
    L.sextans_dist_unique_id.argtypes = [C.c_char_p]
    L.sextans_dist_comm_init.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.sextans_dist_comm_destroy.argtypes = [C.c_void_p]
    L.sextans_dist_spmm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _i32p, C.c_int, C.c_float, C.c_void_p,
                                    C.c_int64, C.c_float, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int,
                                    C.c_void_p]
    if hasattr(raw, "sextans_dist_spmm_rm"):
        L.sextans_dist_spmm_rm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _i32p, C.c_int, C.c_float, C.c_void_p,
                                           C.c_int64, C.c_float, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    pp = C.POINTER(C.c_void_p)
    L.sextans_edges_pack_csc.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f32p, C.POINTER(Edges)]
    L.sextans_edges_free.argtypes = [C.POINTER(Edges)]
    L.sextans_edges_free.restype = None
    L.sextans_edges_decode_csr.argtypes = [_i32p, pp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), pi, pi, pf]
    L.sextans_edges_save.argtypes = [C.c_char_p, C.POINTER(Edges)]
    L.sextans_edges_load.argtypes = [C.c_char_p, C.POINTER(Edges)]
    for fn in ("sextans_chan_b_colsize", "sextans_chan_b_len", "sextans_chan_c_colsize", "sextans_chan_c_len"):
        getattr(L, fn).restype = C.c_int64
    L.sextans_chan_b_colsize.argtypes = [C.c_int, C.c_int]
    L.sextans_chan_b_len.argtypes = [C.c_int, C.c_int, C.c_int]
    L.sextans_chan_c_colsize.argtypes = [C.c_int]
    L.sextans_chan_c_len.argtypes = [C.c_int, C.c_int]
    L.sextans_chan_pack_b.argtypes = [C.c_int, C.c_int, C.c_int, _f32p, pp]
    L.sextans_chan_unpack_b.argtypes = [C.c_int, C.c_int, C.c_int, pp, _f32p]
    L.sextans_chan_pack_c.argtypes = [C.c_int, C.c_int, _f32p, pp]
    L.sextans_chan_unpack_c.argtypes = [C.c_int, C.c_int, pp, _f32p]
    L.sextans_set_matrix_edges.argtypes = [C.c_void_p, _i32p, pp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.sextans_invoke.argtypes = [C.c_void_p, C.c_void_p, pp, pp, C.c_int, pp, pp, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.sextans_device_count.argtypes = [ip]
    L.sextans_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.sextans_destroy.argtypes = [C.c_void_p]
    L.sextans_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.sextans_get_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
    L.sextans_set_matrix_csr.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, _i32p, _i32p,
                                         _f32p]
    L.sextans_set_matrix_csr_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64,
                                                C.c_void_p, C.c_void_p, C.c_void_p]
    L.sextans_spmm_host.argtypes = [C.c_void_p, C.c_int, C.c_float, _f32p, C.c_float, _f32p,
                                    C.c_int, C.POINTER(C.c_double)]
    L.sextans_spmm_device.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int64,
                                      C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.sextans_spmm_device2.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int64, C.c_float,
                                       C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    if hasattr(raw, "sextans_spmm_device_rm"):      # (absent from the older builds tools/nasa_ab.py compares against)
        L.sextans_spmm_device_rm.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int64, C.c_float,
                                             C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    L.sextans_spmm_device_rows.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int64, C.c_float,
                                           C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p]
    L.sextans_spmm_csr.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _i32p, _i32p,
                                   _f32p, _f32p, C.c_float, _f32p]
    L.sextans_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                       C.POINTER(C.c_double)]
    L.sextans_profile_read_post.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.sextans_profile_reset.argtypes = [C.c_void_p]
    L.sextans_phase_timing_read.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.sextans_last_kernel.restype = C.c_char_p
    L.sextans_last_kernel.argtypes = [C.c_void_p]
    L.sextans_gen_csr_host.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                       pi, pi, pf, C.POINTER(C.c_int64)]
    L.sextans_gen_csr_device.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_uint64,
                                         C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_int64)]
    L.sextans_gen_stencil2d_host.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, pi, pi, pf,
                                             C.POINTER(C.c_int64)]
    L.sextans_gen_stencil2d_device.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                               C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_int64)]
    L.sextans_gen_kkt_host.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, pi, pi, pf, C.POINTER(C.c_int64)]
    L.sextans_gen_kkt_device.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.sextans_gen_fem3d_host.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                         pi, pi, pf, C.POINTER(C.c_int64)]
    L.sextans_gen_fem3d_device.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                           C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_int64)]
    L.sextans_gen_powerlaw_host.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                            pi, pi, pf, C.POINTER(C.c_int64)]
    L.sextans_gen_powerlaw_device.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                              C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
    L.sextans_set_matrix_bell.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _i32p, u16p]
    L.sextans_set_matrix_bell_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.sextans_spmm_bell_device.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int64, C.c_float,
                                           C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    if hasattr(raw, "sextans_dist_spmm_bell"):
        L.sextans_spmm_bell_device2.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int64, C.c_float,
                                                C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
        L.sextans_dist_spmm_bell.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _i32p, C.c_int, C.c_float, C.c_void_p,
                                             C.c_int64, C.c_float, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    L.sextans_gen_bell_banded_host.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, pi, C.POINTER(C.POINTER(C.c_uint16))]
    L.sextans_gen_bell_banded_device.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_void_p),
                                                 C.POINTER(C.c_void_p)]
    L.sextans_gen_bell_host.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, pi,
                                        C.POINTER(C.POINTER(C.c_uint16))]
    L.sextans_gen_bell_device.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.sextans_gen_uniform_bf16_host.argtypes = [u16p, C.c_int64, C.c_uint64]
    L.sextans_gen_uniform_bf16_device.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p]
    L.sextans_gen_uniform_host.argtypes = [_f32p, C.c_int64, C.c_uint64]
    L.sextans_gen_uniform_device.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_uint64,
                                             C.c_void_p]
    L.sextans_device_free.argtypes = [C.c_int, C.c_void_p]
    L.sextans_device_alloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
    L.sextans_device_copy.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.sextans_prepare.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.sextans_dist_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _i32p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.sextans_dist_bind_library.argtypes = [C.c_char_p]
    _lib = raw
    return raw


def _check(rc, where):
    if rc != 0:
        raise SextansError(rc, where)


def _buf(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a if a.size else np.zeros(1, dtype)


def _take(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


# ------------------------------------------------------------------ L2: host sparse library

def read_suitsparse_matrix(path, fmt=FMT_CSR, cache=None):
    """-> (ptr, idx, val, M, K, nnz); raises SextansError where the reference would exit(1).
    cache: None = parse the text; True = go through the binary container next to the file
    (path + ".csr.sxbin" / ".csc.sxbin"); a string = that container path."""
    L = lib()
    M, K, nnz = C.c_int(), C.c_int(), C.c_int()
    p, i, v = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)()
    if cache is None:
        _check(L.sextans_mtx_read(os.fsencode(path), fmt, M, K, nnz, p, i, v), f"mtx_read({path})")
    else:
        hit = C.c_int()
        cpath = None if cache is True else os.fsencode(cache)
        _check(L.sextans_mtx_read_cached(os.fsencode(path), cpath, fmt, M, K, nnz, p, i, v, C.byref(hit)),
               f"mtx_read_cached({path})")
        read_suitsparse_matrix.last_cache_hit = bool(hit.value)
    n_ptr = (M.value if fmt == FMT_CSR else K.value) + 1
    out = (_take(p, n_ptr, np.int32), _take(i, nnz.value, np.int32),
           _take(v, nnz.value, np.float32), M.value, K.value, nnz.value)
    for q in (p, i, v):
        L.sextans_host_free(q)
    return out


def matrix_save(path, fmt, M, K, ptr, idx, val):
    _check(lib().sextans_matrix_save(os.fsencode(path), fmt, M, K, int(len(idx)), _buf(ptr, np.int32),
                                     _buf(idx, np.int32), _buf(val, np.float32)), f"matrix_save({path})")


def matrix_load(path):
    """-> (fmt, ptr, idx, val, M, K, nnz)"""
    L = lib()
    fmt, M, K, nnz = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    p, i, v = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)()
    _check(L.sextans_matrix_load(os.fsencode(path), fmt, M, K, nnz, p, i, v), f"matrix_load({path})")
    n_ptr = (M.value if fmt.value == FMT_CSR else K.value) + 1
    out = (fmt.value, _take(p, n_ptr, np.int32), _take(i, nnz.value, np.int32), _take(v, nnz.value, np.float32),
           M.value, K.value, nnz.value)
    for q in (p, i, v):
        L.sextans_host_free(q)
    return out


def CSC_2_CSR(M, K, NNZ, csc_col_ptr, csc_row_idx, csc_val):
    rp = np.zeros(M + 1, np.int32)
    ci = np.zeros(max(NNZ, 1), np.int32)
    cv = np.zeros(max(NNZ, 1), np.float32)
    _check(lib().sextans_csc_to_csr(M, K, NNZ, _buf(csc_col_ptr, np.int32),
                                    _buf(csc_row_idx, np.int32), _buf(csc_val, np.float32),
                                    rp, ci, cv), "csc_to_csr")
    return rp, ci[:NNZ].copy(), cv[:NNZ].copy()


def pack_csr(M, K, row_ptr, col_idx, val, lanes_per_row=4, min_reuse_x100=400):
    """Build the packed row-bucketed form (host); returns a dict of numpy arrays + scalars."""
    L = lib()
    P = Packed()
    _check(L.sextans_pack_csr(M, K, _buf(row_ptr, np.int32), _buf(col_idx, np.int32), _buf(val, np.float32),
                              lanes_per_row, min_reuse_x100, C.byref(P)), "pack_csr")
    try:
        nb, sl = P.nblk, P.stream_len
        out = dict(M=P.M, K=P.K, nnz=P.nnz, lanes_per_row=P.lanes_per_row, nblk=nb, stream_len=sl,
                   max_dict=P.max_dict, nnz_in_panel_blocks=P.nnz_in_panel_blocks,
                   blk_row=_take(P.blk_row, nb + 1, np.int32), dict_ptr=_take(P.dict_ptr, nb + 1, np.int32),
                   row_off=_take(P.row_off, M + 1, np.int32), idx16=_take(P.idx16, sl, np.uint16),
                   col32=_take(P.col32, sl, np.int32), val=_take(P.val, sl, np.float32))
        out["dict"] = _take(P.dict, int(out["dict_ptr"][-1]), np.int32)
        # decode through the C decoder as well
        ci = np.zeros(max(int(P.nnz), 1), np.int32)
        va = np.zeros(max(int(P.nnz), 1), np.float32)
        _check(L.sextans_unpack_csr(C.byref(P), _buf(row_ptr, np.int32), ci, va), "unpack_csr")
        out["decoded_col_idx"], out["decoded_val"] = ci[:P.nnz], va[:P.nnz]
        return out
    finally:
        L.sextans_packed_free(C.byref(P))


def window_pack_csr(M, K, row_ptr, col_idx, val, rows_per_wave=319, window_cols=65536):
    """Build the K-windowed stream of the accumulator-resident kernel (host); returns a dict:
    wave_step0[nwaves+1], val[steps*32] (float32), row[steps*32] (local row; == rows_per_wave for padding),
    col[steps*32]."""
    L = lib()
    P = WindowPacked()
    _check(L.sextans_window_pack_csr(M, K, _buf(row_ptr, np.int32), _buf(col_idx, np.int32),
                                     _buf(val, np.float32), rows_per_wave, window_cols, C.byref(P)),
           "window_pack_csr")
    try:
        n = int(P.steps) * 32
        words = _take(P.stream, n, np.uint64)
        hi = (words >> np.uint64(32)).astype(np.uint32)
        return dict(M=P.M, K=P.K, nnz=P.nnz, rows_per_wave=P.rows_per_wave, window_cols=P.window_cols,
                    nwaves=P.nwaves, steps=int(P.steps), padded_lower_bound=int(P.padded_lower_bound),
                    wave_step0=_take(P.wave_step0, P.nwaves + 1, np.int32),
                    val=(words & np.uint64(0xffffffff)).astype(np.uint32).view(np.float32),
                    row=(hi >> np.uint32(23)).astype(np.int32), col=(hi & np.uint32(0x7fffff)).astype(np.int32))
    finally:
        L.sextans_window_packed_free(C.byref(P))


# ---- the accelerator's own buffer formats ----

def _rows(a2d):
    """(void *)[n] over the rows of a C-contiguous 2-D array (one pointer per channel)."""
    n = a2d.shape[0]
    arr = (C.c_void_p * n)()
    for i in range(n):
        arr[i] = a2d[i].ctypes.data
    return arr


def _edges_to_dict(E):
    ch = np.stack([_take(E.channel[c], E.chan_len, np.uint64) if E.chan_len else np.zeros(0, np.uint64)
                   for c in range(8)])
    return dict(M=E.M, K=E.K, num_windows=E.num_windows, num_a_len=E.num_a_len, nnz=E.nnz,
                edge_list_ptr=_take(E.edge_list_ptr, E.ptr_len, np.int32), channels=ch)


def edges_pack_csc(M, K, col_ptr, row_idx, val):
    """generate_edge_list_for_all_PEs + edge_list_64bit + the edge_list_ptr padding
    (sextans-host.cpp:114-146): -> dict(edge_list_ptr[ptr_len], channels[8, chan_len] uint64, ...)."""
    L = lib()
    E = Edges()
    _check(L.sextans_edges_pack_csc(M, K, int(len(row_idx)), _buf(col_ptr, np.int32), _buf(row_idx, np.int32),
                                    _buf(val, np.float32), C.byref(E)), "edges_pack_csc")
    try:
        return _edges_to_dict(E)
    finally:
        L.sextans_edges_free(C.byref(E))


def edges_decode_csr(edge_list_ptr, channels, num_windows, M, K):
    """-> (row_ptr, col_idx, val): each row's entries in stream order."""
    L = lib()
    ch = np.ascontiguousarray(channels, np.uint64)
    if ch.shape[1] == 0:
        ch = np.zeros((8, 1), np.uint64)
    nnz = C.c_int64()
    p, i, v = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)()
    _check(L.sextans_edges_decode_csr(_buf(edge_list_ptr, np.int32), _rows(ch), num_windows, M, K, C.byref(nnz),
                                      p, i, v), "edges_decode_csr")
    out = (_take(p, M + 1, np.int32), _take(i, nnz.value, np.int32), _take(v, nnz.value, np.float32))
    for q in (p, i, v):
        L.sextans_host_free(q)
    return out


def edges_save(path, e):
    """Write the container file from a dict as returned by edges_pack_csc."""
    L = lib()
    ch = np.ascontiguousarray(e["channels"], np.uint64)
    ptr = np.ascontiguousarray(e["edge_list_ptr"], np.int32)
    E = Edges(M=e["M"], K=e["K"], num_windows=e["num_windows"], num_a_len=e["num_a_len"], nnz=e["nnz"],
              ptr_len=ptr.size, chan_len=ch.shape[1])
    E.edge_list_ptr = ptr.ctypes.data_as(C.POINTER(C.c_int32))
    for c in range(8):
        E.channel[c] = ch[c].ctypes.data_as(C.POINTER(C.c_uint64))
    _check(L.sextans_edges_save(os.fsencode(path), C.byref(E)), f"edges_save({path})")


def edges_load(path):
    L = lib()
    E = Edges()
    _check(L.sextans_edges_load(os.fsencode(path), C.byref(E)), f"edges_load({path})")
    try:
        return _edges_to_dict(E)
    finally:
        L.sextans_edges_free(C.byref(E))


def chan_pack_b(K, N, B, num_ch_b=4):
    """mat_B_cpu (column major K x N) -> mat_B_fpga_vec[num_ch_b][len] (sextans-host.cpp:152-177)."""
    L = lib()
    ch = np.zeros((num_ch_b, L.sextans_chan_b_len(K, N, num_ch_b)), np.float32)
    _check(L.sextans_chan_pack_b(K, N, num_ch_b, _buf(B, np.float32), _rows(ch)), "chan_pack_b")
    return ch


def chan_unpack_b(K, N, channels):
    ch = np.ascontiguousarray(channels, np.float32)
    B = np.zeros(K * N, np.float32)
    _check(lib().sextans_chan_unpack_b(K, N, ch.shape[0], _rows(ch), _buf(B, np.float32)), "chan_unpack_b")
    return B


def chan_pack_c(M, N, Cm):
    """mat_C_cpu (column major M x N) -> mat_C_fpga_in[8][len] (sextans-host.cpp:179-195)."""
    L = lib()
    ch = np.zeros((8, L.sextans_chan_c_len(M, N)), np.float32)
    _check(L.sextans_chan_pack_c(M, N, _buf(Cm, np.float32), _rows(ch)), "chan_pack_c")
    return ch


def chan_unpack_c(M, N, channels):
    """mat_C_fpga_vec[8][len] -> column major M x N (the indexing of sextans-host.cpp:264-270)."""
    ch = np.ascontiguousarray(channels, np.float32)
    Cm = np.zeros(M * N, np.float32)
    out = _buf(Cm, np.float32)
    _check(lib().sextans_chan_unpack_c(M, N, _rows(ch), out), "chan_unpack_c")
    return out[:M * N]


def init_dense_B(K, N):
    B = np.empty(K * N, np.float32)
    lib().sextans_init_dense_B(K, N, _buf(B, np.float32) if B.size == 0 else B)
    return B


def init_dense_C(M, N):
    Cm = np.empty(M * N, np.float32)
    lib().sextans_init_dense_C(M, N, _buf(Cm, np.float32) if Cm.size == 0 else Cm)
    return Cm


def round_up_n(N):
    return lib().sextans_round_up_n(N)


def verify(M, N, c_cpu, c_dev):
    pct = C.c_float()
    n = lib().sextans_verify(M, N, _buf(c_cpu, np.float32), _buf(c_dev, np.float32), C.byref(pct))
    return n, pct.value


def gflops(M, N, nnz, seconds):
    return lib().sextans_gflops(M, N, nnz, seconds)


def device_count():
    n = C.c_int()
    rc = lib().sextans_device_count(C.byref(n))
    return 0 if rc else n.value


# ------------------------------------------------------------------ L1: the engine

class Engine:
    """One engine per HIP device; upload A once, launch SpMM many times."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        self.device = device
        _check(lib().sextans_create(C.byref(self._h), device), "sextans_create")
        self.M = self.K = self.nnz = 0

    def close(self):
        if self._h:
            lib().sextans_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_option(self, key, value):
        _check(lib().sextans_set_option(self._h, key.encode(), int(value)), f"set_option({key})")

    def dist_spmm(self, comm, world, rank, ranges, N, alpha, d_B, ldb, beta, d_C_in, ldc_in, d_C_out, ldc, nchunks=4,
                  stream=None):
        """Native multi-GPU SpMM (sextans_dist_spmm): this engine holds rows ranges[rank] of the matrix."""
        rr = np.ascontiguousarray(np.asarray(ranges, np.int32).reshape(-1))
        _check(lib().sextans_dist_spmm(self._h, comm, world, rank, rr, N, alpha, d_B, ldb, beta, d_C_in, ldc_in,
                                       d_C_out, ldc, nchunks, stream), "dist_spmm")

    def dist_spmm_rm(self, comm, world, rank, ranges, N, alpha, d_B, ldb, beta, d_C_in, ldc_in, d_C_out, ldc, stream=None):
        """Native multi-GPU SpMM on ROW-major operands (sextans_dist_spmm_rm): the rank's rows are written in place, the exchange is an
        in-place all-gather of contiguous row runs."""
        rr = np.ascontiguousarray(np.asarray(ranges, np.int32).reshape(-1))
        _check(lib().sextans_dist_spmm_rm(self._h, comm, world, rank, rr, N, alpha, d_B, ldb, beta, d_C_in, ldc_in, d_C_out, ldc, stream),
               "dist_spmm_rm")

    def prepare(self, N, rowmajor=False, stream=None):
        """Build everything the first SpMM call for N columns would build inside itself (sextans_prepare)."""
        _check(lib().sextans_prepare(self._h, N, 1 if rowmajor else 0, stream), "sextans_prepare")

    def dist_prepare(self, comm, world, rank, ranges, N, nchunks=4, form=0, stream=None):
        """Collective: exchanges, plan build, tables and workspaces of the dist entry point `form` (0 sextans_dist_spmm, 1 _rm, 2 _bell)
        outside the timed call; OK on all ranks or an error on all ranks (sextans_dist_prepare)."""
        rr = np.ascontiguousarray(np.asarray(ranges, np.int32).reshape(-1))
        _check(lib().sextans_dist_prepare(self._h, comm, world, rank, rr, N, nchunks, form, stream),
               "sextans_dist_prepare")

    def export_plan(self, lanes_per_row=4):
        """The packed row-bucketed form the engine built ON THE DEVICE for the current matrix, read back in the layout of
        pack_csr (the host builder of the same format): dict of numpy arrays + scalars."""
        P = Packed()
        _check(lib().sextans_export_plan(self._h, lanes_per_row, C.byref(P)), "export_plan")
        try:
            nb, sl, M = P.nblk, P.stream_len, P.M
            out = dict(M=P.M, K=P.K, nnz=P.nnz, lanes_per_row=P.lanes_per_row, nblk=nb, stream_len=sl,
                       max_dict=P.max_dict, nnz_in_panel_blocks=P.nnz_in_panel_blocks,
                       blk_row=_take(P.blk_row, nb + 1, np.int32), dict_ptr=_take(P.dict_ptr, nb + 1, np.int32),
                       row_off=_take(P.row_off, M + 1, np.int32), idx16=_take(P.idx16, sl, np.uint16),
                       col32=_take(P.col32, sl, np.int32), val=_take(P.val, sl, np.float32))
            out["dict"] = _take(P.dict, int(out["dict_ptr"][-1]), np.int32)
            return out
        finally:
            lib().sextans_packed_free(C.byref(P))

    def export_row_order(self):
        """(order, kind): order[i] = row at position i of the clustered plan (identity when kind == 0); see include/sextans_amd.h."""
        L = lib()
        L.sextans_export_row_order.argtypes = [C.c_void_p, _i32p, C.POINTER(C.c_int)]
        order, kind = np.empty(max(self.M, 1), np.int32), C.c_int()
        _check(L.sextans_export_row_order(self._h, order, C.byref(kind)), "export_row_order")
        return order[:self.M], kind.value

    def reassociated_rows(self):
        """Hub rows whose sums are formed in pieces under the current "split_rows" setting (ascending)."""
        n = C.c_int()
        _check(lib().sextans_reassociated_rows(self._h, np.zeros(1, np.int32), 0, C.byref(n)), "reassociated_rows")
        rows = np.zeros(max(n.value, 1), np.int32)
        _check(lib().sextans_reassociated_rows(self._h, rows, n.value, C.byref(n)), "reassociated_rows")
        return rows[:n.value]

    def align_row(self, N, row):
        """Largest row <= `row` where a row-range call keeps the whole-matrix kernel (sextans_align_row)."""
        out = C.c_int()
        _check(lib().sextans_align_row(self._h, N, int(row), C.byref(out)), "align_row")
        return out.value

    def get_stat(self, key):
        v = C.c_double()
        _check(lib().sextans_get_stat(self._h, key.encode(), C.byref(v)), f"get_stat({key})")
        return v.value

    def get_option(self, key):
        v = C.c_int64()
        _check(lib().sextans_get_option(self._h, key.encode(), C.byref(v)), f"get_option({key})")
        return v.value

    def set_matrix_csr(self, M, K, row_ptr, col_idx, val):
        nnz = int(np.asarray(col_idx).shape[0])
        _check(lib().sextans_set_matrix_csr(self._h, M, K, nnz, _buf(row_ptr, np.int32),
                                            _buf(col_idx, np.int32), _buf(val, np.float32)),
               "set_matrix_csr")
        self.M, self.K, self.nnz = M, K, nnz

    def set_matrix_csr_device(self, M, K, nnz, d_row_ptr, d_col_idx, d_val):
        _check(lib().sextans_set_matrix_csr_device(self._h, M, K, nnz, d_row_ptr, d_col_idx,
                                                   d_val), "set_matrix_csr_device")
        self.M, self.K, self.nnz = M, K, nnz

    def spmm(self, N, alpha, B, beta, C_inout, rp_time=1):
        """Host buffers, C updated in place (cpu_spmm_CSR semantics).  Returns device ns for all
        rp_time repeats (what tapa::invoke returns, sextans-host.cpp:237)."""
        B = np.ascontiguousarray(B, np.float32)
        if not (isinstance(C_inout, np.ndarray) and C_inout.dtype == np.float32
                and C_inout.flags.c_contiguous):
            raise TypeError("C must be a contiguous float32 numpy array (updated in place)")
        if B.size != self.K * N or C_inout.size != self.M * N:
            raise ValueError("B must hold K*N and C must hold M*N floats")
        ns = C.c_double()
        _check(lib().sextans_spmm_host(self._h, N, alpha, _buf(B, np.float32), beta,
                                       C_inout if C_inout.size else np.zeros(1, np.float32),
                                       rp_time, C.byref(ns)), "spmm_host")
        return ns.value

    def set_matrix_edges(self, edge_list_ptr, channels, NUM_ITE, NUM_A_LEN, M, K):
        ch = np.ascontiguousarray(channels, np.uint64)
        _check(lib().sextans_set_matrix_edges(self._h, _buf(edge_list_ptr, np.int32), _rows(ch), NUM_ITE,
                                              NUM_A_LEN, M, K), "set_matrix_edges")
        self.M, self.K = M, K

    def invoke(self, edge_list_ptr, edge_list_ch, mat_B_ch, mat_C_ch_in, NUM_ITE, NUM_A_LEN, M, K, P_N,
               alpha_u, beta_u, mat_C_ch=None):
        """tapa::invoke(Sextans, ...)'s argument list (sextans-host.cpp:237-251) on numpy buffers.
        edge_list_ptr=None reuses the matrix already set.  -> (mat_C_ch[8, len], elapsed_ns)."""
        bch = np.ascontiguousarray(mat_B_ch, np.float32)
        cin = np.ascontiguousarray(mat_C_ch_in, np.float32)
        cout = np.zeros_like(cin) if mat_C_ch is None else mat_C_ch
        ns = C.c_double()
        if edge_list_ptr is None:
            ptr, ach = None, (C.c_void_p * 8)()
        else:
            ptr_arr = _buf(edge_list_ptr, np.int32)
            ach_arr = np.ascontiguousarray(edge_list_ch, np.uint64)
            ptr, ach = ptr_arr.ctypes.data, _rows(ach_arr)
        _check(lib().sextans_invoke(self._h, ptr, ach, _rows(bch), bch.shape[0], _rows(cin), _rows(cout),
                                    NUM_ITE, NUM_A_LEN, M, K, P_N, alpha_u, beta_u, C.byref(ns)), "invoke")
        self.M, self.K = M, K
        return cout, ns.value

    def spmm_device(self, N, alpha, d_B, ldb, beta, d_C_in, d_C_out, ldc, stream=None):
        _check(lib().sextans_spmm_device(self._h, N, alpha, d_B, ldb, beta, d_C_in, d_C_out, ldc,
                                         stream), "spmm_device")

    # ---- blocked-ELL bf16 path (BASELINE config 5)
    def set_matrix_bell(self, M, K, ell_width, block_col, block_val_bf16):
        _check(lib().sextans_set_matrix_bell(self._h, M, K, ell_width, _buf(block_col, np.int32),
                                             _buf(block_val_bf16, np.uint16)), "set_matrix_bell")

    def set_matrix_bell_device(self, M, K, ell_width, d_block_col, d_block_val):
        _check(lib().sextans_set_matrix_bell_device(self._h, M, K, ell_width, d_block_col, d_block_val),
               "set_matrix_bell_device")

    def spmm_bell_device(self, N, alpha, d_B_bf16, ldb, beta, d_C_in, d_C_out, ldc, stream=None):
        _check(lib().sextans_spmm_bell_device(self._h, N, alpha, d_B_bf16, ldb, beta, d_C_in, d_C_out, ldc,
                                              stream), "spmm_bell_device")

    def spmm_bell_device2(self, N, alpha, d_B_bf16, ldb, beta, d_C_in, ldc_in, d_C_out, ldc, stream=None):
        _check(lib().sextans_spmm_bell_device2(self._h, N, alpha, d_B_bf16, ldb, beta, d_C_in, ldc_in, d_C_out, ldc, stream), "spmm_bell_device2")

    def dist_spmm_bell(self, comm, world, rank, ranges, N, alpha, d_B_bf16, ldb, beta, d_C_in, ldc_in, d_C_out, ldc, stream=None):
        """Blocked-ELL over several GPUs (sextans_dist_spmm_bell): this engine holds the block rows of ranges[rank]."""
        rr = np.ascontiguousarray(np.asarray(ranges, np.int32).reshape(-1))
        _check(lib().sextans_dist_spmm_bell(self._h, comm, world, rank, rr, N, alpha, d_B_bf16, ldb, beta, d_C_in, ldc_in, d_C_out, ldc, stream),
               "dist_spmm_bell")

    def spmm_device2(self, N, alpha, d_B, ldb, beta, d_C_in, ldc_in, d_C_out, ldc_out, stream=None):
        _check(lib().sextans_spmm_device2(self._h, N, alpha, d_B, ldb, beta, d_C_in, ldc_in, d_C_out,
                                          ldc_out, stream), "spmm_device2")

    def spmm_device_rm(self, N, alpha, d_B, ldb, beta, d_C_in, ldc_in, d_C_out, ldc_out, stream=None):
        """ROW-major operands (B[k * ldb + n], C[m * ldc + n]): no layout pass on the LDS-panel paths (include/sextans_amd.h)."""
        _check(lib().sextans_spmm_device_rm(self._h, N, alpha, d_B, ldb, beta, d_C_in, ldc_in, d_C_out, ldc_out, stream),
               "spmm_device_rm")

    def spmm_device_rows(self, N, alpha, d_B, ldb, beta, d_C_in, ldc_in, d_C_out, ldc_out, row_begin, row_end,
                         reuse_b_panels=False, stream=None):
        _check(lib().sextans_spmm_device_rows(self._h, N, alpha, d_B, ldb, beta, d_C_in, ldc_in, d_C_out,
                                              ldc_out, row_begin, row_end, 1 if reuse_b_panels else 0,
                                              stream), "spmm_device_rows")

    def profile_read(self):
        k, n, r = C.c_double(), C.c_int64(), C.c_double()
        _check(lib().sextans_profile_read(self._h, C.byref(k), C.byref(n), C.byref(r)),
               "profile_read")
        return k.value, n.value, r.value

    def profile_read_post(self):
        """(mean ns, launches) of what calls launched behind their SpMM kernel (the reordered form's C staging -> C pass)."""
        k, n = C.c_double(), C.c_int64()
        _check(lib().sextans_profile_read_post(self._h, C.byref(k), C.byref(n)), "profile_read_post")
        return k.value, n.value

    def phase_timing_read(self):
        out = (C.c_int64 * 8)()
        _check(lib().sextans_phase_timing_read(self._h, out), "phase_timing_read")
        return list(out)

    def profile_reset(self):
        _check(lib().sextans_profile_reset(self._h), "profile_reset")

    def last_kernel(self):
        return lib().sextans_last_kernel(self._h).decode()


def spmm_csr(M, N, K, NNZ, ALPHA, CSRRowPtr, CSRColIndex, CSRVal, mat_B, BETA, mat_C):
    """One-shot call with cpu_spmm_CSR's argument list (sparse_helper.h:262-272); in place."""
    _check(lib().sextans_spmm_csr(M, N, K, NNZ, ALPHA, _buf(CSRRowPtr, np.int32),
                                  _buf(CSRColIndex, np.int32), _buf(CSRVal, np.float32),
                                  _buf(mat_B, np.float32), BETA, mat_C), "spmm_csr")
    return mat_C


# ------------------------------------------------------------------ synthetic inputs

def gen_csr_host(M, K, mean_nnz, seed, r0=0, r1=None, bandwidth=0):
    L = lib()
    r1 = M if r1 is None else r1
    p, i, v = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)()
    nnz = C.c_int64()
    _check(L.sextans_gen_csr_host(M, K, mean_nnz, bandwidth, seed, r0, r1, p, i, v, C.byref(nnz)),
           "gen_csr_host")
    out = (_take(p, r1 - r0 + 1, np.int32), _take(i, nnz.value, np.int32),
           _take(v, nnz.value, np.float32))
    for q in (p, i, v):
        L.sextans_host_free(q)
    return out


def gen_csr_device(device, M, K, mean_nnz, seed, r0=0, r1=None, bandwidth=0):
    """-> (d_row_ptr, d_col_idx, d_val, nnz) raw device pointers (ints); free with device_free."""
    r1 = M if r1 is None else r1
    p, i, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nnz = C.c_int64()
    _check(lib().sextans_gen_csr_device(device, M, K, mean_nnz, bandwidth, seed, r0, r1, C.byref(p),
                                        C.byref(i), C.byref(v), C.byref(nnz)), "gen_csr_device")
    return p.value, i.value, v.value, nnz.value


def gen_fem3d_host(nx, ny, nz, dof, seed, r0=0, r1=None):
    L = lib()
    r1 = nx * ny * nz * dof if r1 is None else r1
    p, i, v = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)()
    nnz = C.c_int64()
    _check(L.sextans_gen_fem3d_host(nx, ny, nz, dof, seed, r0, r1, p, i, v, C.byref(nnz)), "gen_fem3d_host")
    out = (_take(p, r1 - r0 + 1, np.int32), _take(i, nnz.value, np.int32), _take(v, nnz.value, np.float32))
    for q in (p, i, v):
        L.sextans_host_free(q)
    return out


def device_alloc(device, nbytes):
    """hipMalloc through the library's own HIP runtime (free with device_free)."""
    ptr = C.c_void_p()
    _check(lib().sextans_device_alloc(device, nbytes, C.byref(ptr)), "device_alloc")
    return ptr.value


COPY_H2D, COPY_D2H, COPY_D2D = 1, 2, 3


def device_copy(device, dst, src, nbytes, kind=COPY_D2D):
    """Synchronous copy through the library's own HIP runtime (never a second dlopen of libamdhip64 by another name: where torch
    bundles its own copy that maps a second runtime, which fails on this one's pointers).  dst / src: integer addresses."""
    _check(lib().sextans_device_copy(device, C.c_void_p(dst), C.c_void_p(src), nbytes, kind), "device_copy")


def upload_csr(device, rp, ci, v):
    """Host CSR arrays -> device copies (torch allocations would be freed with their tensors; these are hipMalloc'ed through the
    library and freed with device_free).  Returns device pointers (rp, ci, v)."""
    out = []
    for arr, dt in ((rp, np.int32), (ci, np.int32), (v, np.float32)):
        a = np.ascontiguousarray(arr, dt)
        ptr = device_alloc(device, a.nbytes)
        if a.nbytes:
            device_copy(device, ptr, a.ctypes.data, a.nbytes, COPY_H2D)
        out.append(ptr)
    return tuple(out)


def permute_symmetric_device(device, M, nnz, d_rp, d_ci, d_v, new_of_old):
    """P A P^T on the device (row / column i -> new_of_old[i]); returns new device pointers (rp, ci, v) -- free with device_free."""
    p, i, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
    no = np.ascontiguousarray(new_of_old, np.int32)
    L = lib()
    L.sextans_csr_permute_symmetric_device.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                       C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    _check(L.sextans_csr_permute_symmetric_device(device, M, nnz, d_rp, d_ci, d_v, no.ctypes.data, C.byref(p), C.byref(i), C.byref(v)),
           "csr_permute_symmetric_device")
    return p.value, i.value, v.value


def slice_rows_device(device, r0, r1, d_rp, d_ci, d_v):
    """Rows [r0, r1) of a device CSR matrix as (rp, ci, v, nnz): rp is a new device array (device_free), ci / v point INTO the originals."""
    p, first, nnz = C.c_void_p(), C.c_int64(), C.c_int64()
    L = lib()
    L.sextans_csr_slice_rows_device.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    _check(L.sextans_csr_slice_rows_device(device, r0, r1, d_rp, C.byref(p), C.byref(first), C.byref(nnz)), "csr_slice_rows_device")
    return p.value, d_ci + 4 * first.value, d_v + 4 * first.value, nnz.value


def gen_fem3d_device(device, nx, ny, nz, dof, seed, r0=0, r1=None):
    r1 = nx * ny * nz * dof if r1 is None else r1
    p, i, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nnz = C.c_int64()
    _check(lib().sextans_gen_fem3d_device(device, nx, ny, nz, dof, seed, r0, r1, C.byref(p), C.byref(i),
                                          C.byref(v), C.byref(nnz)), "gen_fem3d_device")
    return p.value, i.value, v.value, nnz.value


def _gen_host(fn, args, nrows):
    L = lib()
    p, i, v = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)()
    nnz = C.c_int64()
    _check(fn(*args, p, i, v, C.byref(nnz)), fn.__name__)
    out = (_take(p, nrows + 1, np.int32), _take(i, nnz.value, np.int32), _take(v, nnz.value, np.float32))
    for q in (p, i, v):
        L.sextans_host_free(q)
    return out


def _gen_device(fn, args):
    p, i, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nnz = C.c_int64()
    _check(fn(*args, C.byref(p), C.byref(i), C.byref(v), C.byref(nnz)), fn.__name__)
    return p.value, i.value, v.value, nnz.value


def gen_stencil2d_host(nx, ny, points, dof, seed, r0=0, r1=None):
    """2-D stencil (5 or 9 points) on an nx x ny grid, dof unknowns per node -> (row_ptr, col_idx, val)."""
    r1 = nx * ny * dof if r1 is None else r1
    return _gen_host(lib().sextans_gen_stencil2d_host, (nx, ny, points, dof, seed, r0, r1), r1 - r0)


def gen_stencil2d_device(device, nx, ny, points, dof, seed, r0=0, r1=None):
    r1 = nx * ny * dof if r1 is None else r1
    return _gen_device(lib().sextans_gen_stencil2d_device, (device, nx, ny, points, dof, seed, r0, r1))


def kkt_rows(n, arrow):
    return n + n // 2 + arrow


def gen_kkt_host(n, arrow, seed, r0=0, r1=None):
    """KKT / arrow block structure (n variables, n/2 constraints, `arrow` border rows/columns) -> (row_ptr, col_idx, val)."""
    r1 = kkt_rows(n, arrow) if r1 is None else r1
    return _gen_host(lib().sextans_gen_kkt_host, (n, arrow, seed, r0, r1), r1 - r0)


def gen_kkt_device(device, n, arrow, seed, r0=0, r1=None):
    r1 = kkt_rows(n, arrow) if r1 is None else r1
    return _gen_device(lib().sextans_gen_kkt_device, (device, n, arrow, seed, r0, r1))


def gen_kron_host(n, p_rp, p_ci, pk, variant, seed, r0=0, r1=None):
    """kron(T_n, P) holdout class (include/sextans_amd.h) -> (row_ptr, col_idx, val, K)."""
    L = lib()
    p_rp = np.ascontiguousarray(p_rp, np.int32); p_ci = np.ascontiguousarray(p_ci, np.int32)
    pm = len(p_rp) - 1
    r1 = n * pm if r1 is None else r1
    L.sextans_gen_kron_host.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                        C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_float)),
                                        C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    p, i, v = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)()
    nnz, K = C.c_int64(), C.c_int()
    _check(L.sextans_gen_kron_host(n, pm, pk, p_rp, p_ci, variant, seed, r0, r1, C.byref(p), C.byref(i), C.byref(v), C.byref(nnz),
                                   C.byref(K)), "gen_kron_host")
    out = (_take(p, r1 - r0 + 1, np.int32), _take(i, nnz.value, np.int32), _take(v, nnz.value, np.float32), K.value)
    for q in (p, i, v):
        L.sextans_host_free(q)
    return out


def gen_kron_device(device, n, p_rp, p_ci, pk, variant, seed, r0=0, r1=None):
    """Device form -> (d_row_ptr, d_col_idx, d_val, nnz, K); free the arrays with device_free."""
    L = lib()
    p_rp = np.ascontiguousarray(p_rp, np.int32); p_ci = np.ascontiguousarray(p_ci, np.int32)
    pm = len(p_rp) - 1
    r1 = n * pm if r1 is None else r1
    L.sextans_gen_kron_device.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int)]
    p, i, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nnz, K = C.c_int64(), C.c_int()
    _check(L.sextans_gen_kron_device(device, n, pm, pk, p_rp, p_ci, variant, seed, r0, r1, C.byref(p), C.byref(i), C.byref(v),
                                     C.byref(nnz), C.byref(K)), "gen_kron_device")
    return p.value, i.value, v.value, nnz.value, K.value


def gen_powerlaw_host(M, K, xmin, tail_x100, max_len, seed, r0=0, r1=None):
    """Power-law row lengths, P(len >= x) = (xmin/x)^(tail_x100/100) on [xmin, max_len] -> (row_ptr, col_idx, val)."""
    L = lib()
    r1 = M if r1 is None else r1
    p, i, v = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)()
    nnz = C.c_int64()
    _check(L.sextans_gen_powerlaw_host(M, K, xmin, tail_x100, max_len, seed, r0, r1, p, i, v, C.byref(nnz)),
           "gen_powerlaw_host")
    out = (_take(p, r1 - r0 + 1, np.int32), _take(i, nnz.value, np.int32), _take(v, nnz.value, np.float32))
    for q in (p, i, v):
        L.sextans_host_free(q)
    return out


def gen_powerlaw_device(device, M, K, xmin, tail_x100, max_len, seed, r0=0, r1=None):
    r1 = M if r1 is None else r1
    p, i, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nnz = C.c_int64()
    _check(lib().sextans_gen_powerlaw_device(device, M, K, xmin, tail_x100, max_len, seed, r0, r1, C.byref(p),
                                             C.byref(i), C.byref(v), C.byref(nnz)), "gen_powerlaw_device")
    return p.value, i.value, v.value, nnz.value


def gen_bell_host(M, K, ell_width, seed):
    L = lib()
    c, v = C.POINTER(C.c_int)(), C.POINTER(C.c_uint16)()
    _check(L.sextans_gen_bell_host(M, K, ell_width, seed, c, v), "gen_bell_host")
    n = (M // 32) * ell_width
    out = (_take(c, n, np.int32), np.ctypeslib.as_array(v, shape=(n * 1024,)).astype(np.uint16, copy=True))
    L.sextans_host_free(c); L.sextans_host_free(v)
    return out


def gen_bell_device(device, M, K, ell_width, seed):
    c, v = C.c_void_p(), C.c_void_p()
    _check(lib().sextans_gen_bell_device(device, M, K, ell_width, seed, C.byref(c), C.byref(v)), "gen_bell_device")
    return c.value, v.value


def gen_bell_banded_host(M, K, half_width, seed):
    """Block-banded blocked-ELL (2 * half_width + 1 consecutive block columns per block row): block rows share columns."""
    L = lib()
    c, v = C.POINTER(C.c_int)(), C.POINTER(C.c_uint16)()
    _check(L.sextans_gen_bell_banded_host(M, K, half_width, seed, c, v), "gen_bell_banded_host")
    n = (M // 32) * (2 * half_width + 1)
    out = (_take(c, n, np.int32), np.ctypeslib.as_array(v, shape=(n * 1024,)).astype(np.uint16, copy=True))
    L.sextans_host_free(c); L.sextans_host_free(v)
    return out


def gen_bell_banded_device(device, M, K, half_width, seed):
    c, v = C.c_void_p(), C.c_void_p()
    _check(lib().sextans_gen_bell_banded_device(device, M, K, half_width, seed, C.byref(c), C.byref(v)), "gen_bell_banded_device")
    return c.value, v.value


def gen_uniform_bf16_host(n, seed):
    a = np.empty(max(n, 1), np.uint16)
    _check(lib().sextans_gen_uniform_bf16_host(a, n, seed), "gen_uniform_bf16_host")
    return a[:n]


def gen_uniform_bf16_device(device, d_ptr, n, seed, stream=None):
    _check(lib().sextans_gen_uniform_bf16_device(device, d_ptr, n, seed, stream), "gen_uniform_bf16_device")


def gen_uniform_host(n, seed):
    a = np.empty(max(n, 1), np.float32)
    _check(lib().sextans_gen_uniform_host(a, n, seed), "gen_uniform_host")
    return a[:n]


def gen_uniform_device(device, d_ptr, n, seed, stream=None):
    _check(lib().sextans_gen_uniform_device(device, d_ptr, n, seed, stream), "gen_uniform_device")


def partition_rows_by_nnz(row_ptr, world):
    """C-ABI twin of sextans_amd.dist.partition_rows_by_nnz: [(r0, r1)] * world."""
    rp = _buf(row_ptr, np.int32)
    out = np.zeros(2 * world, np.int32)
    _check(lib().sextans_partition_rows_by_nnz(len(rp) - 1, rp, world, out), "partition_rows_by_nnz")
    return [(int(out[2 * g]), int(out[2 * g + 1])) for g in range(world)]


def dist_bind_library(path=None):
    """Which library the collectives come from (None: the default RCCL search); sextans_dist_bind_library."""
    _check(lib().sextans_dist_bind_library(path.encode() if path else None), "sextans_dist_bind_library")


def dist_unique_id():
    """128-byte RCCL id (rank 0 creates it, the other ranks receive it by the caller's own means)."""
    buf = C.create_string_buffer(128)
    _check(lib().sextans_dist_unique_id(buf), "dist_unique_id")
    return buf.raw


def dist_comm_init(device, world, rank, uid):
    comm = C.c_void_p()
    _check(lib().sextans_dist_comm_init(C.byref(comm), device, world, rank, uid), "dist_comm_init")
    return comm


def dist_comm_destroy(comm):
    _check(lib().sextans_dist_comm_destroy(comm), "dist_comm_destroy")


def device_free(device, d_ptr):
    _check(lib().sextans_device_free(device, d_ptr), "device_free")
