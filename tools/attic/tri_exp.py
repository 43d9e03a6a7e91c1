import os, sys, time
sys.path.insert(0, ".")
import numpy as np, scipy.sparse as sp, torch
from sextans_amd import api, meshgen
st = torch.cuda.current_stream().cuda_stream
rp, ci, v = api.gen_fem3d_host(64, 64, 64, 3, 3); Mf = 64 ** 3 * 3
t = meshgen.permute_symmetric(rp, ci, v, Mf, meshgen.node_permutation(Mf // 3, 3, 2))
L = sp.tril(sp.csr_matrix((t[2], t[1], t[0]), shape=(Mf, Mf)), format="csr"); L.sort_indices()
rp, ci, v = L.indptr.astype(np.int32), L.indices.astype(np.int32), L.data.astype(np.float32)
M = Mf; nnz = int(rp[-1]); N = 16
B = torch.empty(M * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
api.gen_uniform_device(0, B.data_ptr(), M * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
for rc in (-1, 2, 0):
    e = api.Engine(0); e.set_option("row_cluster", rc); e.set_matrix_csr(M, M, rp, ci, v)
    f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), M, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): f()
    torch.cuda.synchronize(); w = (time.time() - t0) / 20
    g = e.get_stat
    print(f"rc={rc}: {w*1e6:.1f} us/step {e.last_kernel()} state={int(g('row_cluster'))} decline={int(g('cluster_decline'))} natural {g('panel_rows_natural')/1e6:.2f} M clustered {g('panel_rows_clustered')/1e6:.2f} M piece_rows {int(g('piece_path_rows'))}", flush=True)
    e.close()
