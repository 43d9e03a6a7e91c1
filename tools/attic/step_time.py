#!/usr/bin/env python3
"""tools/step_time.py <workload> [key=value ...] -- wall time per back-to-back spmm_device step (repack + kernel launches)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from sextans_amd import api  # noqa: E402
from sweep import workload  # noqa: E402
name = sys.argv[1]
opts = dict(kv.split("=") for kv in sys.argv[2:] if "=" in kv)
iters = int(opts.pop("iters", 1000))
w = workload(name); M, K, N = w["M"], w["K"], w["N"]
e = api.Engine(0)
if "host" in w: e.set_matrix_csr(M, K, *w["host"]); nnz = w["nnz"]
else: p = api.gen_fem3d_device(0, *w["fem"]); nnz = p[3]; e.set_matrix_csr_device(M, K, nnz, *p[:3])
for k, v in opts.items(): e.set_option(k, int(v))
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
B = torch.rand(K * N, device=dev); Cin = torch.rand(M * N, device=dev); Cout = torch.empty(M * N, device=dev)
f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
for _ in range(10): f()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters): f()
torch.cuda.synchronize(); per = (time.perf_counter() - t0) / iters
print(f"{name} {opts} {e.last_kernel()}: {per*1e6:.2f} us/step  {api.gflops(M, N, nnz, per):.1f} GFLOP/s")
