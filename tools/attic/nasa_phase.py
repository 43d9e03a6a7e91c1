"""Where a wavefront's life goes in the small-matrix case (VERDICT r02 task 5): nasa4704 N=16 and the config-3 stand-in, per kernel form.
Engine option `phase_timing`: one workgroup in 16 (v2) / 128 (round-1 kernel) adds wavefront-0's cycle counts per phase."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api
import bench
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
NAMES = {"v2": ["args+meta+extents+dictionary (round trip 1)", "row entries + first panel + barrier (round trip 2)", "row loops (all tiles)",
                "drains, C_in wait, C stores, panel turnover"],
         "v1": ["meta, row extents, first entries", "dictionary -> B rows -> LDS, barrier", "row streaming / compute", "C tile + epilogue"]}

def phases(e, M, K, N, launches=400):
    B = torch.empty(K * N, device=dev); Cin = torch.empty(M * N, device=dev); Cout = torch.empty(M * N, device=dev)
    api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
    f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
    f(); torch.cuda.synchronize()
    e.set_option("phase_timing", 1)
    for _ in range(launches):
        f()
    torch.cuda.synchronize()
    t = e.phase_timing_read()
    e.set_option("phase_timing", 0)
    return t, e.last_kernel()

def report(tag, e, M, K, N, nnz):
    out = bench._measure(api, torch, e, M, K, N, nnz, dev, st, 2000)
    t, kern = phases(e, M, K, N)
    n = max(t[4], 1)
    total = sum(t[:4]) / n
    ghz = (sum(t[:4]) / max(t[5], 1)) / 10.0 if t[5] else 0.0   # cycles per 10 ns tick
    names = NAMES["v2" if kern.startswith("spmm_csr_panel_v2") else "v1"]
    print(f"{tag}: kernel={kern} eager {out['us_per_step']} us/step; sampled wavefronts {n}; wave life {total:.0f} cycles = "
          f"{total / (ghz * 1e3) if ghz else 0:.2f} us at {ghz:.2f} GHz (clock64 / wall_clock64)")
    for i in range(4):
        print(f"    {t[i] / n:8.0f} cycles  {100.0 * t[i] / max(sum(t[:4]), 1):5.1f} %  {names[i]}")

rp, ci, v, M, K, nnz = api.read_suitsparse_matrix(os.path.join(ROOT, "matrices", "nasa4704", "nasa4704.mtx"))
for opts in ({}, {"small_v2": 0}, {"panel_v2": 0}):
    e = api.Engine(0)
    for k, val in opts.items():
        e.set_option(k, val)
    e.set_matrix_csr(M, K, rp, ci, v)
    Bh, Ch = api.init_dense_B(K, 16), api.init_dense_C(M, 16)
    e.spmm(16, 0.85, Bh, -2.06, Ch.copy(), rp_time=10)
    ns = e.spmm(16, 0.85, Bh, -2.06, Ch, rp_time=2000)
    print(f"nasa4704 N=16 {opts}: rp_time loop (hipGraph) {ns / 2000 / 1e3:.3f} us per repeat ({e.last_kernel()})")
    report(f"nasa4704 N=16 {opts}", e, M, K, 16, nnz)
    e.close()
p, i, vv, nz = api.gen_fem3d_device(0, 35, 19, 7, 3, 3)
for opts in ({}, {"fuse_b": 0}):
    e = api.Engine(0)
    for k, val in opts.items():
        e.set_option(k, val)
    e.set_matrix_csr_device(13965, 13965, nz, p, i, vv)
    report(f"config-3 stand-in N=128 {opts}", e, 13965, 13965, 128, nz)
    e.close()
p, i, vv, nz = api.gen_fem3d_device(0, 110, 110, 110, 3, 3)
e = api.Engine(0)
e.set_matrix_csr_device(3993000, 3993000, nz, p, i, vv)
report("FEM 4M N=16 (throughput case, for scale)", e, 3993000, 3993000, 16, nz)
e.close()
