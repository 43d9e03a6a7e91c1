#!/usr/bin/env python3
"""tools/refine_exp.py <synth spec> <N> "k=v,k=v" ... -- plan figures + kernel time of the reordered form under option sets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api, sweep
M, K, p, i, v, nnz = sweep._synth(sys.argv[1], 0)
N = int(sys.argv[2])
st = torch.cuda.current_stream().cuda_stream
B = torch.empty(K * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
for a in sys.argv[3:]:
    o = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv)
    e = api.Engine(0)
    for k, val in o.items(): e.set_option(k, val)
    e.set_matrix_csr_device(M, K, nnz, p, i, v)
    f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
    for _ in range(3): f()
    e.set_option("profile", 1); e.profile_reset()
    for _ in range(10): f()
    torch.cuda.synchronize()
    k_ns, n, r_ns = e.profile_read()
    print(f"{o}: kernel {k_ns/1e3:.1f} us  panel rows {e.get_stat('panel_rows_clustered')/1e6:.2f} M  blocks {e.get_stat('panel_blocks_clustered'):.0f}  plan {e.get_stat('plan_build_s'):.3f} s", flush=True)
    e.close()
