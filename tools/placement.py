"""Does the placement of the operands change the FEM N = 16 kernel time?  B, C_in, C_out at different byte offsets inside larger allocations (same box, same engine, same plan)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api
st = torch.cuda.current_stream().cuda_stream
M = 110 ** 3 * 3; N = 16
p = api.gen_fem3d_device(0, 110, 110, 110, 3, 3)
e = api.Engine(0); e.set_matrix_csr_device(M, M, p[3], *p[:3])
pad = 1 << 22   # floats
big = [torch.empty(M * N + pad, device="cuda") for _ in range(3)]
for t in big: api.gen_uniform_device(0, t.data_ptr(), M * N + pad, 41, st)
def run(ob, oi, oo):
    B, Cin, Cout = big[0].data_ptr() + 4 * ob, big[1].data_ptr() + 4 * oi, big[2].data_ptr() + 4 * oo
    f = lambda: e.spmm_device(N, 0.85, B, M, -2.06, Cin, Cout, M, st)
    for _ in range(3): f()
    e.set_option("profile", 1); e.profile_reset()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(30): f()
    torch.cuda.synchronize(); w = (time.time() - t0) / 30
    k, _, r = e.profile_read(); e.set_option("profile", 0)
    return k / 1e3, r / 1e3, w * 1e6
print("base addresses mod 2 MiB:", [t.data_ptr() % (1 << 21) for t in big])
for offs in ((0, 0, 0), (0, 0, 1024), (0, 1024, 2048), (0, 16384, 32768), (0, 64 * 1024, 128 * 1024), (0, 1 << 19, 1 << 20), (32, 32, 32), (0, 0, 16), (0, 16, 0), (0, 0, 0)):
    k, r, w = run(*offs)
    print(f"offsets (floats) B {offs[0]:8d} C_in {offs[1]:8d} C_out {offs[2]:8d}: kernel {k:6.1f} us repack {r:5.1f} wall {w:6.1f}", flush=True)
