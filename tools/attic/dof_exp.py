"""FEM 27-point matrices with 1 .. 12 unknowns per node (rows of 27 .. 324 entries), N = 16 / 64: step time, fraction of 8 TB/s on algorithmic bytes, plan figures -- looking for cliffs between the classes the bench covers."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api
st = torch.cuda.current_stream().cuda_stream
for n, dof in ((125, 2), (90, 4), (70, 6), (50, 12)):
    M = n * n * n * dof
    p = api.gen_fem3d_device(0, n, n, n, dof, 3)
    nnz = p[3]
    e = api.Engine(0); e.set_matrix_csr_device(M, M, nnz, *p[:3])
    for N in (16, 64):
        B = torch.empty(M * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
        api.gen_uniform_device(0, B.data_ptr(), M * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
        f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), M, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20): f()
        torch.cuda.synchronize(); w = (time.time() - t0) / 20
        e.set_option("profile", 1); e.profile_reset()
        for _ in range(10): f()
        torch.cuda.synchronize(); k_ns, _, r_ns = e.profile_read(); e.set_option("profile", 0)
        by = 8 * nnz + 4 * (M + 1) + 12 * M * N
        g = e.get_stat
        print(f"fem {n}^3 x {dof} dof: M={M} nnz={nnz} ({nnz / M:.0f}/row) N={N}: step {w * 1e6:.0f} us = {by / w / 8e12:.3f}, kernel {k_ns / 1e3:.0f} us = {by / (k_ns * 1e-9) / 8e12:.3f} "
              f"({e.last_kernel()}, row_cluster {int(g('row_cluster'))}, blocks {int(g('panel_blocks_clustered') or g('panel_blocks'))}, "
              f"idx/val {g('index_stream_entries') / max(g('value_stream_entries'), 1):.2f})", flush=True)
        del B, Cin, Cout
    e.close()
    for q in p[:3]: api.device_free(0, q)
    torch.cuda.empty_cache()
