#!/usr/bin/env python3
"""tools/run_reordered.py [N=16] [iters=5] [numbering=random] -- the 4M-row 3-dof FEM matrix under a node renumbering, a few steps
(for rocprofv3 kernel-trace / --pmc passes of the reordered form)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api, sweep
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
num = sys.argv[3] if len(sys.argv) > 3 else "random"
M, K, p, i, v, nnz = sweep._synth(f"synth:femperm:110:110:110:3:{num}", 0)
e = api.Engine(0)
e.set_matrix_csr_device(M, K, nnz, p, i, v)
st = torch.cuda.current_stream().cuda_stream
B = torch.empty(K * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
for _ in range(iters):
    e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
torch.cuda.synchronize()
print(e.last_kernel(), "row_cluster", e.get_stat("row_cluster"), "plan_build_s", e.get_stat("plan_build_s"))
