"""Is the wide kernel at N=128 bound by B-panel traffic?  Same nnz/row, different z-plane footprints."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api
import bench
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
for dims in ((30, 30, 440, 3), (60, 60, 110, 3), (110, 110, 33, 3), (110, 110, 110, 3)):
    for N, opts in ((128, {"cols_per_lane": 8}), (128, {"cols_per_lane": 4}), (16, {})):
        nx, ny, nz, dof = dims
        M = K = nx * ny * nz * dof
        p, i, v, nnz = api.gen_fem3d_device(0, nx, ny, nz, dof, 3)
        e = api.Engine(0)
        for k, val in opts.items():
            e.set_option(k, val)
        e.set_matrix_csr_device(M, K, nnz, p, i, v)
        out = bench._measure(api, torch, e, M, K, N, nnz, dev, st, 20)
        print(dims, N, opts, out["kernel"], "kernel_us", out["kernel_us"], "ps/nnz/col", round(out["kernel_us"] * 1e6 / nnz / N, 3), "frac", out["roofline_frac_kernel"], flush=True)
        e.close()
        for q in (p, i, v):
            api.device_free(0, q)
        torch.cuda.empty_cache()
