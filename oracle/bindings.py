"""ctypes bindings for the CPU checkers.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (sextans_amd/) never does.

  Oracle   -- oracle/liboracle.so, our plain-C restatement (sextans_oracle.c)
  Ref      -- oracle/_ref/libsextans_ref.so, the reference's own host functions
              (sparse_helper.h / mmio.h compiled by oracle/Makefile); may be absent.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")

CSR, CSC = 0, 1


def build(quiet=True):
    """(Re)build liboracle.so and, when /root/reference is mounted, _ref/."""
    out = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


def _take(ptr, n, dtype):
    """Copy n elements out of a malloc'ed C array into a fresh numpy array."""
    if n == 0:
        return np.zeros(0, dtype=dtype)
    arr = np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)
    return arr


class Oracle:
    def __init__(self):
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = self.lib = C.CDLL(path)
        L.orc_read_suitsparse_matrix.restype = C.c_int
        L.orc_read_suitsparse_matrix.argtypes = [
            C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
            C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_int)),
            C.POINTER(C.POINTER(C.c_float))]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_csc_to_csr.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f32p, _i32p, _i32p,
                                     _f32p]
        L.orc_cpu_spmm_csr.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _i32p, _i32p,
                                       _f32p, _f32p, C.c_float, _f32p]
        L.orc_cpu_spmm_csr_rows.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                            _i32p, _i32p, _f32p, _f32p, C.c_float, _f32p]
        L.orc_cpu_spmm_csr_fma.restype = None
        L.orc_cpu_spmm_csr_fma.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, _i32p, _i32p, _f32p, _f32p, C.c_float, _f32p]
        L.orc_time_spmm_rows.restype = C.c_double
        L.orc_time_spmm_rows.argtypes = L.orc_cpu_spmm_csr_rows.argtypes
        L.orc_time_spmm_csr_omp.restype = C.c_double
        L.orc_time_spmm_csr_omp.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, _i32p, _i32p, _f32p, _f32p,
                                            C.c_float, _f32p, C.POINTER(C.c_int)]
        L.orc_init_B.argtypes = [C.c_int, C.c_int, _f32p]
        L.orc_init_C.argtypes = [C.c_int, C.c_int, _f32p]
        L.orc_verify.restype = C.c_int
        L.orc_verify.argtypes = [C.c_int, C.c_int, _f32p, _f32p, C.POINTER(C.c_float)]
        u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
        f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
        L.orc_bell_spmm.restype = None
        L.orc_bell_spmm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _i32p, u16p, u16p, C.c_float,
                                    C.c_float, _f32p, f64p]
        L.orc_edge_stream.restype = C.c_int
        L.orc_edge_stream.argtypes = [C.c_int, C.c_int, _i32p, _i32p, _f32p, C.POINTER(C.POINTER(C.c_int)),
                                      C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_long)]
        L.orc_gflops.restype = C.c_double
        L.orc_gflops.argtypes = [C.c_int, C.c_int, C.c_long, C.c_double]

    def read_mtx(self, path, fmt=CSC):
        """-> (err, M, K, nnz, ptr, idx, val); arrays are None when err != 0."""
        M, K, nnz = C.c_int(), C.c_int(), C.c_int()
        p, i, v = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)()
        err = self.lib.orc_read_suitsparse_matrix(os.fsencode(path), fmt, M, K, nnz, p, i, v)
        if err:
            return err, 0, 0, 0, None, None, None
        n_ptr = (M.value if fmt == CSR else K.value) + 1
        out = (0, M.value, K.value, nnz.value, _take(p, n_ptr, np.int32),
               _take(i, nnz.value, np.int32), _take(v, nnz.value, np.float32))
        for q in (p, i, v):
            self.lib.orc_free(q)
        return out

    def csc_to_csr(self, M, K, col_ptr, row_idx, val):
        nnz = int(row_idx.shape[0])
        rp = np.zeros(M + 1, np.int32)
        ci = np.zeros(max(nnz, 1), np.int32)[:nnz].copy()
        cv = np.zeros(max(nnz, 1), np.float32)[:nnz].copy()
        self.lib.orc_csc_to_csr(M, K, nnz, np.ascontiguousarray(col_ptr), _pad(row_idx),
                                _pad(val), rp, _pad_out(ci), _pad_out(cv))
        return rp, ci, cv

    def load_csr(self, path):
        """The reference host's sequence (sextans-host.cpp:67-84): CSC read, then CSC->CSR."""
        err, M, K, nnz, cp, ri, cv = self.read_mtx(path, CSC)
        if err:
            raise RuntimeError(f"oracle loader error {err} on {path}")
        rp, ci, v = self.csc_to_csr(M, K, cp, ri, cv)
        return M, K, nnz, rp, ci, v

    def spmm(self, M, N, K, alpha, row_ptr, col_idx, val, B, beta, C_inout):
        """In place on C_inout (column-major M x N, flat float32)."""
        self.lib.orc_cpu_spmm_csr(M, N, K, int(col_idx.shape[0]), alpha, row_ptr, _pad(col_idx),
                                  _pad(val), B, beta, C_inout)
        return C_inout

    def spmm_rows(self, r0, r1, M, N, K, alpha, row_ptr, col_idx, val, B, beta, C_inout):
        self.lib.orc_cpu_spmm_csr_rows(r0, r1, M, N, K, alpha, row_ptr, _pad(col_idx), _pad(val),
                                       B, beta, C_inout)
        return C_inout

    def time_spmm_rows(self, r0, r1, M, N, K, alpha, row_ptr, col_idx, val, B, beta, C_inout):
        return self.lib.orc_time_spmm_rows(r0, r1, M, N, K, alpha, row_ptr, _pad(col_idx),
                                           _pad(val), B, beta, C_inout)

    def spmm_fma(self, M, N, K, alpha, row_ptr, col_idx, val, B, beta, C_inout):
        """cpu_spmm_CSR's loop nest with fused multiply-adds (psum = fmaf(a, b, psum), epilogue fmaf(alpha, psum, beta * c)): the CPU
        statement of the engine's "exact" = 0 kernels and of its fp32 matrix-core path, which must agree with it bit for bit."""
        self.lib.orc_cpu_spmm_csr_fma(M, N, K, alpha, row_ptr, _pad(col_idx), _pad(val), B, beta, C_inout)

    def time_spmm_omp(self, M, N, K, alpha, row_ptr, col_idx, val, B, beta, C_inout):
        """cpu_spmm_CSR's loop nest on all host cores (OpenMP); -> (seconds, threads used)."""
        t = C.c_int(0)
        sec = self.lib.orc_time_spmm_csr_omp(M, N, K, alpha, row_ptr, _pad(col_idx), _pad(val), B, beta, C_inout,
                                             C.byref(t))
        return sec, t.value

    def bell_spmm(self, M, K, N, ell_width, block_col, block_val, B_bf16, alpha, beta, C_inout):
        """Blocked-ELL bf16 restatement (config 5; parity unpinned by the reference).  In place on
        C_inout; returns sum|a*b| per output (float64) for condition-aware tolerances."""
        asum = np.zeros(M * N, np.float64)
        self.lib.orc_bell_spmm(M, K, N, ell_width, np.ascontiguousarray(block_col, np.int32),
                               np.ascontiguousarray(block_val, np.uint16),
                               np.ascontiguousarray(B_bf16, np.uint16), alpha, beta, C_inout, asum)
        return asum

    def edge_stream(self, M, K, col_ptr, row_idx, val):
        """Restatement of generate_edge_list_for_all_PEs + edge_list_64bit: -> (ptr[num_windows+1],
        channels[8, chan_len] uint64)."""
        p, ch, n = C.POINTER(C.c_int)(), C.POINTER(C.c_uint64)(), C.c_long()
        err = self.lib.orc_edge_stream(M, K, np.ascontiguousarray(col_ptr, np.int32), _pad(np.asarray(row_idx, np.int32)),
                                       _pad(np.asarray(val, np.float32)), p, ch, n)
        if err:
            raise RuntimeError(f"orc_edge_stream error {err}")
        nwin = (K + 4095) // 4096
        out = (_take(p, nwin + 1, np.int32),
               (_take(ch, 8 * n.value, np.uint64) if n.value else np.zeros(0, np.uint64)).reshape(8, n.value))
        self.lib.orc_free(p)
        self.lib.orc_free(ch)
        return out

    def init_B(self, K, N):
        B = np.empty(K * N, np.float32)
        self.lib.orc_init_B(K, N, B)
        return B

    def init_C(self, M, N):
        Cm = np.empty(M * N, np.float32)
        self.lib.orc_init_C(M, N, Cm)
        return Cm

    def verify(self, M, N, c_cpu, c_dev):
        pct = C.c_float()
        n = self.lib.orc_verify(M, N, c_cpu, c_dev, C.byref(pct))
        return n, pct.value

    def gflops(self, M, N, nnz, seconds):
        return self.lib.orc_gflops(M, N, nnz, seconds)


def _pad(a):
    """ndpointer rejects 0-length views of some layouts; hand C a valid 1-element buffer."""
    a = np.ascontiguousarray(a)
    return a if a.size else np.zeros(1, a.dtype)


def _pad_out(a):
    return a if a.size else np.zeros(1, a.dtype)


class Ref:
    """The reference's own functions.  Raises FileNotFoundError when _ref/ was never built."""

    def __init__(self):
        path = os.path.join(_HERE, "_ref", "libsextans_ref.so")
        if not os.path.exists(path):
            if os.path.exists("/root/reference/src/sparse_helper.h"):
                build()
            else:
                raise FileNotFoundError(path)
        L = self.lib = C.CDLL(path)
        pi, pf = C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_float))
        L.ref_load_mtx.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int), pi, pi, pf, pi, pi, pf]
        L.ref_load_mtx_csr.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       C.POINTER(C.c_int), pi, pi, pf]
        L.ref_free.argtypes = [C.c_void_p]
        L.ref_cpu_spmm_csr.restype = C.c_double
        L.ref_cpu_spmm_csr.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _i32p, _i32p,
                                       _f32p, _f32p, C.c_float, _f32p]

        L.ref_generate_edge_list.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f32p,
                                             C.POINTER(C.c_int), C.POINTER(C.c_int), pi, pi, pi, pf]

    def generate_edge_list(self, M, K, col_ptr, row_idx, val):
        """The reference's generate_edge_list_for_all_PEs (64 PEs, window 4096, distance 10).
        -> (ptr[num_windows+1], row[64, L], col[64, L], val[64, L]); row == -1 marks a bubble."""
        nw, Ln = C.c_int(), C.c_int()
        p, r, c, v = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)()
        self.lib.ref_generate_edge_list(M, K, int(len(row_idx)), np.ascontiguousarray(col_ptr, np.int32),
                                        _pad(np.asarray(row_idx, np.int32)), _pad(np.asarray(val, np.float32)),
                                        nw, Ln, p, r, c, v)
        n = 64 * Ln.value
        out = (_take(p, nw.value + 1, np.int32), _take(r, n, np.int32).reshape(64, Ln.value),
               _take(c, n, np.int32).reshape(64, Ln.value), _take(v, n, np.float32).reshape(64, Ln.value))
        for q in (p, r, c, v):
            self.lib.ref_free(q)
        return out

    @staticmethod
    def available():
        return os.path.exists(os.path.join(_HERE, "_ref", "libsextans_ref.so")) or \
            os.path.exists("/root/reference/src/sparse_helper.h")

    def load(self, path):
        """-> dict(M,K,nnz, csc=(ptr,idx,val), csr=(ptr,idx,val)) as sextans-host.cpp:67-84."""
        M, K, nnz = C.c_int(), C.c_int(), C.c_int()
        a = [C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)(),
             C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)()]
        self.lib.ref_load_mtx(os.fsencode(path), M, K, nnz, *a)
        m, k, n = M.value, K.value, nnz.value
        out = dict(M=m, K=k, nnz=n,
                   csc=(_take(a[0], k + 1, np.int32), _take(a[1], n, np.int32),
                        _take(a[2], n, np.float32)),
                   csr=(_take(a[3], m + 1, np.int32), _take(a[4], n, np.int32),
                        _take(a[5], n, np.float32)))
        for q in a:
            self.lib.ref_free(q)
        return out

    def load_csr_direct(self, path):
        M, K, nnz = C.c_int(), C.c_int(), C.c_int()
        p, i, v = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)()
        self.lib.ref_load_mtx_csr(os.fsencode(path), M, K, nnz, p, i, v)
        out = (M.value, K.value, nnz.value, _take(p, M.value + 1, np.int32),
               _take(i, nnz.value, np.int32), _take(v, nnz.value, np.float32))
        for q in (p, i, v):
            self.lib.ref_free(q)
        return out

    def spmm(self, M, N, K, alpha, row_ptr, col_idx, val, B, beta, C_inout):
        """In place; returns seconds spent inside the reference's cpu_spmm_CSR."""
        return self.lib.ref_cpu_spmm_csr(M, N, K, int(col_idx.shape[0]), alpha, row_ptr,
                                         _pad(col_idx), _pad(val), B, beta, C_inout)
