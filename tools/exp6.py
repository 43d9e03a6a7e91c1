import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api
import bench
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
def run(dims, N, opts, iters):
    nx, ny, nz, dof = dims
    M = K = nx * ny * nz * dof
    p, i, v, nnz = api.gen_fem3d_device(0, nx, ny, nz, dof, 3)
    e = api.Engine(0)
    for k, val in opts.items():
        e.set_option(k, val)
    e.set_matrix_csr_device(M, K, nnz, p, i, v)
    out = bench._measure(api, torch, e, M, K, N, nnz, dev, st, iters)
    print(dims, N, opts, out["kernel"], "step_us", out["us_per_step"], "kernel_us", out["kernel_us"], "frac", out["roofline_frac_kernel"], flush=True)
    e.close()
    for q in (p, i, v):
        api.device_free(0, q)
    torch.cuda.empty_cache()
big = (110, 110, 110, 3)
run(big, 16, {}, 100)
run(big, 16, {"panel_v2": 0}, 100)
for N in (32, 64, 128):
    for tpw in (0, 1, 2, 4):
        if tpw <= N // 16:
            run(big, N, {"tiles_per_wg": tpw}, 20)
run(big, 128, {"tiles_per_wg": 8}, 20)
run(big, 256, {}, 10)
run((160, 160, 160, 1), 16, {}, 50)
run((160, 160, 160, 1), 16, {"panel_v2": 0}, 50)
run((160, 160, 160, 1), 64, {}, 20)
run((160, 160, 160, 1), 64, {"panel_v2": 0}, 20)
