#!/usr/bin/env python3
"""One matrix, one N, fast mode (rows routed to the fp32 matrix cores), a few launches -- for rocprofv3 --kernel-trace / --pmc.
    python tools/rowblock_one.py blocks|fem3|fem6 N [thr]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from rowblock_exp import dense_pattern  # noqa: E402
from sextans_amd import api  # noqa: E402

which, N = sys.argv[1], int(sys.argv[2])
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 30
torch.cuda.set_device(0)
st = torch.cuda.current_stream().cuda_stream
if which == "fem3":
    M = K = 110 ** 3 * 3; p, i, v, nnz = api.gen_fem3d_device(0, 110, 110, 110, 3, 3)
elif which == "fem6":
    M = K = 80 ** 3 * 6; p, i, v, nnz = api.gen_fem3d_device(0, 80, 80, 80, 6, 3)
else:
    prp, pci = dense_pattern(32)
    p, i, v, nnz, K = api.gen_kron_device(0, 32768, prp, pci, 32, 0, 7); M = 32768 * 32
e = api.Engine(0)
for k, val in dict(exact=0, dense_tile_fill_x100=thr, mfma_dense_tiles=2).items():
    e.set_option(k, val)
if len(sys.argv) > 4:   # measurement switch (SEXTANS_DEBUG_OPTIONS=1): tiles of 16 columns per wavefront
    e.set_option("rowblock_tiles", int(sys.argv[4]))
e.set_matrix_csr_device(M, K, nnz, p, i, v)
B = torch.empty(K * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
for _ in range(6):
    e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
torch.cuda.synchronize()
print(e.last_kernel(), e.get_stat("dense_tile_fraction"), "us_per_step", round((time.perf_counter() - t0) / 20 * 1e6, 1))
