"""A last zoo of classes at N = 16 (4M rows unless noted): narrow bands, very sparse random rows, 2-D multi-dof stencils -- sanity of the automatic choices."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api, sweep
st = torch.cuda.current_stream().cuda_stream
N = 16
for spec in ("synth:banded:4000000:40:50", "synth:banded:4000000:8:8", "synth:banded:4000000:20:300", "synth:uniform:4000000:3", "synth:uniform:4000000:10",
             "synth:stencil2d:1100:1100:9:3", "synth:stencil2d:2000:1000:5:2", "synth:fem3d:100:100:100:2", "synth:kkt:1000000:2"):
    M, K, p0, p1, p2, nnz = sweep._synth(spec, 0)
    e = api.Engine(0); e.set_matrix_csr_device(M, K, nnz, p0, p1, p2)
    B = torch.empty(K * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
    api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
    f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): f()
    torch.cuda.synchronize(); w = (time.time() - t0) / 20
    by = 8 * nnz + 4 * (M + 1) + 4 * K * N + 8 * M * N
    g = e.get_stat
    print(f"{spec:34s} M={M:8d} nnz={nnz:10d} ({nnz / M:5.1f}/row): {w * 1e6:8.1f} us frac {by / w / 8e12:.3f} {e.last_kernel()} rc={int(g('row_cluster'))} decline={int(g('cluster_decline'))} colwise={int(g('colwise'))} coherence={g('row_coherence'):.2f}", flush=True)
    e.close(); del B, Cin, Cout
    for q in (p0, p1, p2): api.device_free(0, q)
