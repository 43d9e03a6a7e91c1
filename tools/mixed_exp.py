"""A MIXED packed plan (row blocks with and without B-row reuse) timed on one library build:  python tools/mixed_exp.py <lib.so> [N ...]
Matrix: the rows of a 70^3 x 3-dof FEM matrix (1.03 M rows) alternating, 4096 rows at a time, with rows of uniformly random columns --
the dictionary blocks run out of LDS, the others gather from global memory: spmm_csr_panel<4, EXACT, MIXED = true>."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sextans_amd.api as api
api.LIB_PATH = os.path.join(ROOT, sys.argv[1])
import torch
rp, ci, v = api.gen_fem3d_host(70, 70, 70, 3, 3)
M = K = 70 * 70 * 70 * 3
urp, uci, uv = api.gen_csr_host(M, K, 40.0, 4)
sel = (np.arange(M) // 4096) % 2 == 0                       # True: FEM row, False: random row
lens = np.where(sel, np.diff(rp), np.diff(urp))
nrp = np.zeros(M + 1, np.int32); nrp[1:] = np.cumsum(lens)
nci = np.empty(nrp[-1], np.int32); nv = np.empty(nrp[-1], np.float32)
for src_rp, src_ci, src_v, mask in ((rp, ci, v, sel), (urp, uci, uv, ~sel)):
    rows = np.nonzero(mask)[0]
    idx = np.concatenate([np.arange(src_rp[r], src_rp[r + 1]) for r in rows]) if len(rows) < 5000 else None
    if idx is None:                                         # vectorised gather of whole rows
        starts, ln = src_rp[rows], (src_rp[rows + 1] - src_rp[rows])
        off = np.repeat(starts - np.concatenate([[0], np.cumsum(ln)[:-1]]), ln) + np.arange(ln.sum())
        dst = np.repeat(nrp[rows] - np.concatenate([[0], np.cumsum(ln)[:-1]]), ln) + np.arange(ln.sum())
        nci[dst] = src_ci[off]; nv[dst] = src_v[off]
e = api.Engine(0)
e.set_matrix_csr(M, K, nrp, nci, nv)
st = torch.cuda.current_stream().cuda_stream
res = []
for N in [int(x) for x in sys.argv[2:]] or [16, 64]:
    B = torch.empty(K * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
    api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
    f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
    for _ in range(30): f()
    e.set_option("profile", 1); e.profile_reset()
    for _ in range(100): f()
    torch.cuda.synchronize()
    k_ns, n, r_ns = e.profile_read(); e.set_option("profile", 0)
    res.append(f"N={N}: {k_ns / 1e3:.1f} us ({e.last_kernel()}, panel_fraction {e.get_stat('panel_fraction'):.2f})")
print(sys.argv[1].split("/")[-1], " ".join(res), flush=True)
