"""The 3-dof FEM matrix under numberings real codes produce: node-major (generator), DOF-MAJOR (all x unknowns, then y, then z), z-fastest grid order, red-black (odd-even) node order; N = 16: what the automatic plan choice makes of each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sextans_amd import api
st = torch.cuda.current_stream().cuda_stream
n, dof = 100, 3
nn = n * n * n
M = nn * dof
base = api.gen_fem3d_device(0, n, n, n, dof, 3)
nnz = base[3]
node = np.arange(nn, dtype=np.int64)
ix, iy, iz = node % n, node // n % n, node // (n * n)
def expand(node_new):   # node-major rows for a node renumbering
    return (node_new[:, None] * dof + np.arange(dof)[None, :]).reshape(-1)
rb = np.empty(nn, np.int64); par = (ix + iy + iz) % 2; rb[np.argsort(par, kind="stable")] = np.arange(nn)
perms = {"node-major, x fastest (generator)": None,
         "dof-major (all x unknowns, then y, then z)": (np.arange(dof)[None, :] * nn + node[:, None]).reshape(-1),
         "node-major, z fastest": expand(ix * n * n + iy * n + iz),
         "node-major, red-black node order": expand(rb)}
N = 16
B = torch.empty(M * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
api.gen_uniform_device(0, B.data_ptr(), M * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
by = 8 * nnz + 4 * (M + 1) + 12 * M * N
rcs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "-1").split(",")]
for name, perm in perms.items():
  p = base[:3] if perm is None else api.permute_symmetric_device(0, M, nnz, *base[:3], perm)
  for rc in rcs:
    e = api.Engine(0); e.set_option("row_cluster", rc); e.set_matrix_csr_device(M, M, nnz, *p)
    f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), M, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): f()
    torch.cuda.synchronize(); w = (time.time() - t0) / 20
    e.set_option("profile", 1); e.profile_reset()
    for _ in range(10): f()
    torch.cuda.synchronize(); k_ns, _, r_ns = e.profile_read(); e.set_option("profile", 0)
    g = e.get_stat
    print(f"{name:45s} rc={rc:2d}: step {w * 1e6:6.0f} us = {by / w / 8e12:.3f}, kernel {k_ns / 1e3:6.0f} us = {by / (k_ns * 1e-9) / 8e12:.3f}  {e.last_kernel()} row_cluster={int(g('row_cluster'))} "
          f"decline={int(g('cluster_decline'))} panel rows natural {g('panel_rows_natural') / 1e6:.1f} M clustered {g('panel_rows_clustered') / 1e6:.1f} M plan {g('plan_build_s'):.2f} s", flush=True)
    e.close()
  if perm is not None:
    for q in p: api.device_free(0, q)
