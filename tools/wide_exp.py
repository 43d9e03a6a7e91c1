"""Wide-N panel kernel experiment: FEM 110^3 x 3 dof and the config-3 stand-in across cols_per_lane / tiles_per_wg.
    python tools/wide_exp.py [big|small|all]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api
import bench

which = sys.argv[1] if len(sys.argv) > 1 else "all"
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
    out["opts"] = opts
    e.close()
    for q in (p, i, v):
        api.device_free(0, q)
    torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)


if which in ("small", "all"):
    for opts in ({"cols_per_lane": 4}, {"cols_per_lane": 8}, {"cols_per_lane": 8}, {"cols_per_lane": 8, "fuse_b": 0},
                 {"cols_per_lane": 8, "tiles_per_wg": 2}):
        run((35, 19, 7, 3), 128, opts, 300)
if which in ("big", "all"):
    for N in (32, 128):
        for opts in ({"cols_per_lane": 4}, {"cols_per_lane": 8}, {"cols_per_lane": 8}, {"cols_per_lane": 8, "tiles_per_wg": 1}):
            run((110, 110, 110, 3), N, opts, 20)
    run((110, 110, 110, 3), 64, {"cols_per_lane": 8}, 20)
