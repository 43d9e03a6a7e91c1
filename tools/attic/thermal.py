"""FEM 4M N = 16 run back to back for ~8 s: step time per block of 500 steps (the power-capped kernel slows down as the package warms up; short A/B runs and the first `also` entries of bench.py see the fast end)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api
st = torch.cuda.current_stream().cuda_stream
M = 110 ** 3 * 3; N = 16
p = api.gen_fem3d_device(0, 110, 110, 110, 3, 3)
e = api.Engine(0); e.set_matrix_csr_device(M, M, p[3], *p[:3])
B = torch.empty(M * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
api.gen_uniform_device(0, B.data_ptr(), M * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), M, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
for _ in range(3): f()
torch.cuda.synchronize()
t_start = time.time()
for blk in range(24):
    t0 = time.time()
    for _ in range(500): f()
    torch.cuda.synchronize()
    print(f"t = {time.time() - t_start:5.1f} s: {(time.time() - t0) / 500 * 1e6:6.1f} us per step", flush=True)
