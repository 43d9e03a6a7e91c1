#!/usr/bin/env python3
"""tools/run_powerlaw.py [key=value ...] -- the 1M-row power-law matrix, N = 16, a few steps (for rocprofv3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api
opts = dict(kv.split("=") for kv in sys.argv[1:] if "=" in kv)
iters = int(opts.pop("iters", 5)); N = int(opts.pop("N", 16))
M = K = 1_000_000
p, i, v, nnz = api.gen_powerlaw_device(0, M, K, 6, 120, 400_000, 7)
e = api.Engine(0)
for k, val in opts.items():
    e.set_option(k, int(val))
e.set_matrix_csr_device(M, K, nnz, p, i, v)
st = torch.cuda.current_stream().cuda_stream
B = torch.empty(K * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
for _ in range(2):
    f()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    f()
torch.cuda.synchronize()
print(f"powerlaw N={N} {opts}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms/step, kernel {e.last_kernel()}, chains {int(e.get_stat('exact_chain_rows'))}")
