#!/usr/bin/env python3
"""tools/run_bell.py <M> <half_width|-W> [key=value ...] iters=n -- block-banded (or, with a negative second argument,
uniformly random with ell width W) blocked-ELL bf16 SpMM at N = 256, a few launches (for rocprofv3 / PMC passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api
M = int(sys.argv[1]); hw = int(sys.argv[2])
opts = dict(kv.split("=") for kv in sys.argv[3:] if "=" in kv)
iters = int(opts.pop("iters", 5))
K, N = M, 256
if hw >= 0:
    W = 2 * hw + 1
    dc, dv = api.gen_bell_banded_device(0, M, K, hw, 5)
else:
    W = -hw
    dc, dv = api.gen_bell_device(0, M, K, W, 5)
e = api.Engine(0)
for k, v in opts.items():
    e.set_option(k, int(v))
e.set_matrix_bell_device(M, K, W, dc, dv)
api.device_free(0, dv)
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
B = torch.empty(K * N, dtype=torch.int16, device=dev)
Cin = torch.empty(M * N, dtype=torch.float32, device=dev); Cout = torch.empty(M * N, dtype=torch.float32, device=dev)
api.gen_uniform_bf16_device(0, B.data_ptr(), K * N, 51, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 52, st)
f = lambda: e.spmm_bell_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
f(); torch.cuda.synchronize()
e.set_option("profile", 1); e.profile_reset()
for _ in range(iters):
    f()
torch.cuda.synchronize()
k_ns, _, _ = e.profile_read()
nb = (M // 32) * W
flops = 2.0 * N * (1024.0 * nb + M)
print(f"bell M={M} W={W} {opts} kernel={e.last_kernel()} {k_ns/1e6:.3f} ms {flops/(k_ns*1e-9)/1e12:.1f} TFLOP/s = {flops/(k_ns*1e-9)/2.5e15*100:.1f} % of 2.5 PF")
