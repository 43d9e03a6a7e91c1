"""sextans_amd -- MI355X-native SpMM engine behind the Sextans host call surface.

Only what the hot path needs lives here: csrc/ (HIP kernels + C ABI, see include/sextans_amd.h),
api.py (ctypes mirror of the reference's host interface), dist.py (row-range sharding + RCCL
all-gather of C), build.py.  The CPU oracle in oracle/ is test infrastructure and is never
imported from this package.
"""
from .api import (CSC_2_CSR, Engine, SextansError, device_count, gflops, init_dense_B,  # noqa: F401
                  init_dense_C, read_suitsparse_matrix, round_up_n, spmm_csr, verify)
