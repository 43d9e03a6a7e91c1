import sys, os, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/sextans_amd") else ".")
import numpy as np, torch
from sextans_amd import api, meshgen
n = 80; dof = 3
rp, ci, v = api.gen_fem3d_host(n, n, n, dof, 3)
rp, ci, v = np.array(rp), np.array(ci), np.array(v)
M = n * n * n * dof
def with_long_rows(rp, ci, v, rows, L, seed=1):
    rs = np.random.RandomState(seed)
    lens = np.diff(rp).copy()
    parts_c, parts_v = [], []
    prev = 0
    for r in sorted(rows):
        parts_c.append(ci[rp[prev]:rp[r]]); parts_v.append(v[rp[prev]:rp[r]])
        c = np.union1d(ci[rp[r]:rp[r + 1]], rs.choice(M, L, replace=False)).astype(np.int32)
        parts_c.append(c); parts_v.append(rs.uniform(-1, 1, len(c)).astype(np.float32))
        lens[r] = len(c); prev = r + 1
    parts_c.append(ci[rp[prev]:]); parts_v.append(v[rp[prev]:])
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), np.concatenate(parts_c).astype(np.int32), np.concatenate(parts_v).astype(np.float32)
st = torch.cuda.current_stream().cuda_stream
N = 16
B = torch.empty(M * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
api.gen_uniform_device(0, B.data_ptr(), M * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
for label, mat in (("plain", (rp, ci, v)), ("3 rows of ~1000", with_long_rows(rp, ci, v, [1000, M // 2, M - 5], 1000)), ("1 row of ~5000", with_long_rows(rp, ci, v, [M // 3], 5000))):
    for perm in (False, True):
        a = mat
        if perm:
            a = meshgen.permute_symmetric(*mat, M, meshgen.node_permutation(M // dof, dof, 1))
        e = api.Engine(0); e.set_matrix_csr(M, M, *a)
        f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), M, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20): f()
        torch.cuda.synchronize(); w = (time.time() - t0) / 20
        g = e.get_stat
        print(f"{label:18s} {'random order' if perm else 'grid order  '}: {w*1e6:7.1f} us/step {e.last_kernel()} row_cluster={int(g('row_cluster'))} decline={int(g('cluster_decline'))} piece_rows={int(g('piece_path_rows'))} panel_fraction={g('panel_fraction'):.3f}", flush=True)
        e.close()
