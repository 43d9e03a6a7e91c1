"""3-dof FEM matrices from 50 K to 4 M rows at N = 16 / 64: where the dispatcher's policies switch (column-major staging while B fits the L2s, clustered plans from 4096 rows, ...) -- looking for cliffs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api
st = torch.cuda.current_stream().cuda_stream
order = sys.argv[1] if len(sys.argv) > 1 else "grid"
from sextans_amd import meshgen
for n in (12, 16, 20, 25, 32, 40, 44, 50, 64, 80, 110):
    dof = 3
    M = n * n * n * dof
    p = api.gen_fem3d_device(0, n, n, n, dof, 3)
    nnz = p[3]
    if order == "random":
        q = api.permute_symmetric_device(0, M, nnz, *p[:3], meshgen.node_permutation(M // dof, dof, 1))
        for x in p[:3]: api.device_free(0, x)
        p = q + (nnz,)
    e = api.Engine(0); e.set_matrix_csr_device(M, M, nnz, *p[:3])
    line = f"fem {n}^3 x 3 {order}: M={M:8d} nnz={nnz:10d}"
    for N in (16, 64):
        B = torch.empty(M * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
        api.gen_uniform_device(0, B.data_ptr(), M * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
        f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), M, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
        for _ in range(5): f()
        it = max(20, min(2000, int(2e8 / max(nnz, 1))))
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(it): f()
        torch.cuda.synchronize(); w = (time.time() - t0) / it
        by = 8 * nnz + 4 * (M + 1) + 12 * M * N
        line += f" | N={N}: {w * 1e6:8.1f} us {2 * N * (nnz + M) / w / 1e9:8.0f} GF/s frac {by / w / 8e12:.3f} {e.last_kernel().replace('spmm_csr_', '')} rc={int(e.get_stat('row_cluster'))}"
        del B, Cin, Cout
    print(line, flush=True)
    e.close()
    for q in p[:3]: api.device_free(0, q)
