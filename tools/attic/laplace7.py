"""3-D 7-point Laplacian (the most common PDE matrix: 7 entries per row), 128^3 and 160^3, N = 16 / 64: automatic choice against the lane-per-row kernel (kernel = 4) and the panel kernel (kernel = 2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sp, torch
from sextans_amd import api
st = torch.cuda.current_stream().cuda_stream
for n in (128, 160):
    T = sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="csr", dtype=np.float32)
    I = sp.identity(n, format="csr", dtype=np.float32)
    A = (sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I)).tocsr(); A.sort_indices()
    M = n ** 3; nnz = A.nnz
    rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float32)
    for N in (16, 64):
        B = torch.empty(M * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
        api.gen_uniform_device(0, B.data_ptr(), M * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
        line = f"laplace7 {n}^3 M={M} nnz={nnz} N={N}:"
        for opts in ({}, {"colwise_max_len": 0}):
            e = api.Engine(0)
            for k, val in opts.items(): e.set_option(k, val)
            e.set_matrix_csr(M, M, rp, ci, v)
            f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), M, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
            for _ in range(3): f()
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(20): f()
            torch.cuda.synchronize(); w = (time.time() - t0) / 20
            by = 8 * nnz + 4 * (M + 1) + 12 * M * N
            line += f" | {opts or 'auto'}: {w * 1e6:7.1f} us frac {by / w / 8e12:.3f} ({e.last_kernel().replace('spmm_csr_', '')} rc {int(e.get_stat('row_cluster'))} colwise {int(e.get_stat('colwise'))})"
            e.close()
        print(line, flush=True)
        del B, Cin, Cout
