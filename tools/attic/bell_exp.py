#!/usr/bin/env python3
"""tools/bell_exp.py -- config 5 (blocked-ELL 1M x 1M, W=328, N=256): kernel variants, time per launch (HIP events),
agreement between variants."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sextans_amd import api  # noqa: E402

M = K = int(sys.argv[1]) if len(sys.argv) > 1 else 1_048_576
W, N = 328, 256
if len(sys.argv) > 3:                   # probe: K small enough for B to sit in the L2s (every B tile a hit)
    K, W = int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
dc, dv = api.gen_bell_device(0, M, K, W, 5)
e = api.Engine(0)
e.set_matrix_bell_device(M, K, W, dc, dv)
api.device_free(0, dv)
B = torch.empty(K * N, dtype=torch.int16, device=dev)
Cin = torch.empty(M * N, device=dev)
api.gen_uniform_bf16_device(0, B.data_ptr(), K * N, 51, st)
api.gen_uniform_device(0, Cin.data_ptr(), M * N, 52, st)
nb = (M // 32) * W
flops = 2.0 * N * (1024.0 * nb + M)
outs = []
gens = [int(x) for x in os.environ.get("BELL_GEN", "").split(",") if x]
for wide, gen in [(0, 0), (1, 0)] + [(1, g) for g in gens]:
    e.set_option("bell_wide", wide); e.set_option("bell_generation", gen)
    out = torch.empty(M * N, device=dev)
    f = lambda: e.spmm_bell_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), out.data_ptr(), M, st)
    f(); torch.cuda.synchronize()
    e.set_option("profile", 1); e.profile_reset()
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    k_ns, n, r_ns = e.profile_read()
    e.set_option("profile", 0); e.profile_reset()
    outs.append(out)
    print(f"bell_wide={wide} generation={gen}: kernel {k_ns/1e6:.3f} ms  {flops/(k_ns*1e-9)/1e12:.1f} TFLOP/s  mfma util {flops/(k_ns*1e-9)/2.5e15:.4f}", flush=True)
d = (outs[0] - outs[1]).abs().max().item()
print("max |wide - narrow| =", d, " max |C| =", outs[0].abs().max().item())
print(f"M={M} K={K} W={W} N={N}: blocks {nb}, B = {K * N * 2 / 2**20:.1f} MiB")
