"""Config-3-sized FEM matrices (13 965 rows) and a 100 K-row one in grid and random node order at N = 128 / 16: what small renumbered matrices cost and whether the graph plan would help below the 65 536-row limit of the automatic choice."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sextans_amd import api, meshgen
st = torch.cuda.current_stream().cuda_stream
for dims in ((35, 19, 7, 3), (30, 20, 15, 3)):
    nx, ny, nz, dof = dims
    rp, ci, v = api.gen_fem3d_host(nx, ny, nz, dof, 2)
    M = nx * ny * nz * dof
    for order in ("grid", "random"):
        a = (np.array(rp), np.array(ci), np.array(v)) if order == "grid" else meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M // dof, dof, 1))
        for N in (16, 32, 64, 128):
            B = torch.empty(M * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
            api.gen_uniform_device(0, B.data_ptr(), M * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
            line = f"fem {dims} {order:6s} N={N:3d}:"
            for opts in ({}, {"fuse_b": 0}):
                e = api.Engine(0)
                for k, val in opts.items(): e.set_option(k, val)
                e.set_matrix_csr(M, M, *a)
                f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), M, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
                for _ in range(10): f()
                torch.cuda.synchronize(); t0 = time.time()
                for _ in range(500): f()
                torch.cuda.synchronize(); w = (time.time() - t0) / 500
                line += f" | {opts or 'auto'}: {w * 1e6:6.1f} us ({e.last_kernel().replace('spmm_csr_', '')}, rc {int(e.get_stat('row_cluster'))})"
                e.close()
            print(line, flush=True)
            del B, Cin, Cout
