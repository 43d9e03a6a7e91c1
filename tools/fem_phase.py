"""Phase timing of spmm_csr_panel_v2 on the 4M-row FEM matrix at several N (engine option phase_timing)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
dims = tuple(int(x) for x in sys.argv[2].split("x")) if len(sys.argv) > 2 else (110, 110, 110, 3)
M = K = dims[0] * dims[1] * dims[2] * dims[3]
p = api.gen_fem3d_device(0, *dims, 3)
e = api.Engine(0)
for kv in (sys.argv[3].split(',') if len(sys.argv) > 3 else []):
    e.set_option(kv.split('=')[0], int(kv.split('=')[1]))
e.set_matrix_csr_device(M, K, p[3], *p[:3])
names = ["args+meta+extents+dictionary (round trip 1)", "row entries + first panel + barrier (round trip 2)", "row loops (all tiles)",
         "drains, C_in wait, C stores, panel turnover"]
for N in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,32,128").split(",")]:
    B = torch.empty(K * N, device=dev); Cin = torch.empty(M * N, device=dev); Cout = torch.empty(M * N, device=dev)
    api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
    f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
    f(); torch.cuda.synchronize()
    e.set_option("phase_timing", 1)
    for _ in range(5): f()
    torch.cuda.synchronize()
    t = e.phase_timing_read(); e.set_option("phase_timing", 0)
    n = max(t[4], 1); tot = sum(t[:4]) / n
    ghz = (sum(t[:4]) / max(t[5], 1)) / 10.0
    print(f"FEM {dims} N={N}: {e.last_kernel()} wave life {tot:.0f} cycles = {tot / (ghz * 1e3):.2f} us at {ghz:.2f} GHz ({n} sampled wavefronts); per tile {tot / (N // 16):.0f} cycles")
    for i in range(4):
        print(f"    {t[i] / n:9.0f} cycles  {100.0 * t[i] / max(sum(t[:4]), 1):5.1f} %  {names[i]}   (per tile {t[i] / n / (N // 16):.0f})")
    del B, Cin, Cout
