#!/usr/bin/env python3
"""tools/run_one_opts.py <synth:spec | AxBxCxD> <N> <iters> [k=v ...] [--rm] -- one workload, a few steps, engine options set first
(for rocprofv3 kernel-trace / --pmc passes; measurement switches need SEXTANS_DEBUG_OPTIONS=1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api, sweep
spec, N, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if not spec.startswith("synth:"):
    spec = "synth:fem3d:" + spec.replace("x", ":")
M, K, p, i, v, nnz = sweep._synth(spec, 0)
e = api.Engine(0)
for kv in sys.argv[4:]:
    if "=" in kv:
        e.set_option(kv.split("=")[0], int(kv.split("=")[1]))
e.set_matrix_csr_device(M, K, nnz, p, i, v)
st = torch.cuda.current_stream().cuda_stream
B = torch.empty(K * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
for _ in range(iters):
    if "--rm" in sys.argv:
        e.spmm_device_rm(N, 0.85, B.data_ptr(), N, -2.06, Cin.data_ptr(), N, Cout.data_ptr(), N, st)
    else:
        e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
torch.cuda.synchronize()
print(e.last_kernel(), "row_cluster", e.get_stat("row_cluster"), "plan_build_s", e.get_stat("plan_build_s"))
