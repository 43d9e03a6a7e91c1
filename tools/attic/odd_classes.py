"""Sanity sweep over shapes the other tools do not generate: diagonal, tridiagonal, a 5-point stencil in random order, a FEM matrix with half of its rows emptied, block-diagonal dense blocks; N = 16: step time and fraction of 8 TB/s -- looking for anything an order of magnitude off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sp, torch
from sextans_amd import api, meshgen
st = torch.cuda.current_stream().cuda_stream
rs = np.random.RandomState(1)
def csr(A):
    A = A.tocsr(); A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float32)
M = 2_000_000
cases = {}
cases["diagonal"] = csr(sp.diags([rs.uniform(-1, 1, M)], [0], format="csr", dtype=np.float32))
cases["tridiagonal"] = csr(sp.diags([rs.uniform(-1, 1, M - 1), rs.uniform(-1, 1, M), rs.uniform(-1, 1, M - 1)], [-1, 0, 1], format="csr", dtype=np.float32))
rp, ci, v = api.gen_stencil2d_host(1414, 1414, 5, 1, 3); M5 = 1414 * 1414
cases["5-point stencil, random order"] = meshgen.permute_symmetric(rp, ci, v, M5, rs.permutation(M5))
rp, ci, v = api.gen_fem3d_host(64, 64, 64, 3, 3); Mf = 64 ** 3 * 3
A = sp.csr_matrix((np.asarray(v), np.asarray(ci), np.asarray(rp)), shape=(Mf, Mf))
D = sp.diags([(rs.rand(Mf) < 0.5).astype(np.float32)], [0], format="csr")
cases["FEM 3 dof, half of the rows emptied"] = csr(D @ A)
nb = 40000; bs = 48
blk = sp.block_diag([sp.csr_matrix(rs.uniform(-1, 1, (bs, bs)).astype(np.float32))] * 1, format="csr")
cases["block diagonal, 48 x 48 dense blocks"] = csr(sp.kron(sp.identity(nb, format="csr", dtype=np.float32), blk, format="csr"))
L = sp.tril(A, format="csr")
cases["FEM 3 dof, lower triangle only"] = csr(L)
far = sp.csr_matrix((rs.uniform(-1, 1, 4 * Mf).astype(np.float32), (np.repeat(np.arange(Mf), 4), rs.randint(0, Mf, 4 * Mf))), shape=(Mf, Mf))
far.sum_duplicates()
cases["FEM 3 dof + 4 random far entries per row"] = csr(A + far)
cases["FEM 3 dof, random order, lower triangle"] = meshgen.permute_symmetric(*csr(A), Mf, meshgen.node_permutation(Mf // 3, 3, 2))
t = cases["FEM 3 dof, random order, lower triangle"]
cases["FEM 3 dof, random order, lower triangle"] = csr(sp.tril(sp.csr_matrix((t[2], t[1], t[0]), shape=(Mf, Mf)), format="csr"))
N = 16
for name, (rp, ci, v) in cases.items():
    M = len(rp) - 1; nnz = int(rp[-1])
    e = api.Engine(0); e.set_matrix_csr(M, M, np.asarray(rp), np.asarray(ci), np.asarray(v))
    B = torch.empty(M * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
    api.gen_uniform_device(0, B.data_ptr(), M * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
    f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), M, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): f()
    torch.cuda.synchronize(); w = (time.time() - t0) / 20
    by = 8 * nnz + 4 * (M + 1) + 12 * M * N
    print(f"{name:40s} M={M:8d} nnz={nnz:10d}: {w * 1e6:8.1f} us/step frac {by / w / 8e12:.3f} {e.last_kernel()} rc={int(e.get_stat('row_cluster'))} decline={int(e.get_stat('cluster_decline'))} colwise={int(e.get_stat('colwise'))}", flush=True)
    e.close(); del B, Cin, Cout
