#!/usr/bin/env python3
"""tools/run_one.py <workload> [key=value ...] [--iters n] -- run one engine configuration a few times
(for rocprofv3 / PMC passes).  Workloads as in tools/sweep.py."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from sextans_amd import api  # noqa: E402
from sweep import workload, alg_bytes  # noqa: E402

name = sys.argv[1]
opts = dict(kv.split("=") for kv in sys.argv[2:] if "=" in kv and not kv.startswith("--"))
iters = int(opts.pop("iters", 5))
w = workload(name)
M, K, N = w["M"], w["K"], w["N"]
e = api.Engine(0)
ptrs = None
if "host" in w:
    e.set_matrix_csr(M, K, *w["host"]); nnz = w["nnz"]
elif "fem" in w:
    ptrs = api.gen_fem3d_device(0, *w["fem"]); nnz = ptrs[3]; e.set_matrix_csr_device(M, K, nnz, *ptrs[:3])
else:
    mean, bw, seed = w["gen"]
    ptrs = api.gen_csr_device(0, M, K, mean, seed, bandwidth=bw); nnz = ptrs[3]
    e.set_matrix_csr_device(M, K, nnz, *ptrs[:3])
for k, v in opts.items():
    e.set_option(k, int(v))
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
B = torch.empty(K * N, device=dev); Cin = torch.empty(M * N, device=dev); Cout = torch.empty(M * N, device=dev)
api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
for _ in range(2):
    e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
e.set_option("profile", 1); e.profile_reset()
for _ in range(iters):
    e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
torch.cuda.synchronize()
k_ns, n, r_ns = e.profile_read()
by = alg_bytes(M, K, N, nnz)
if opts.get("phase_timing") == "1":
    pt = e.phase_timing_read()
    print("  phase cycles/wave: prologue %.0f staging %.0f stream %.0f epilogue %.0f (sampled waves %d)" % tuple([x / max(pt[4], 1) for x in pt[:4]] + [pt[4]]))
    print("  wave lifetime %.2f us (100 MHz clock) => clock64 rate %.0f MHz" % (pt[5] / max(pt[4], 1) / 100.0, sum(pt[:4]) / max(pt[5], 1) * 100.0))
print(f"{name} {opts} kernel={e.last_kernel()} {k_ns/1e3:.2f} us repack {r_ns/1e3:.2f} us alg {by/(k_ns*1e-9)/1e9:.1f} GB/s nnz={nnz}")
