"""Small canonical runs: nasa4704 N=16 (eager + rp_time loop) and the config-3 stand-in N=128, per option set."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api
import bench
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
rp, ci, v, M, K, nnz = api.read_suitsparse_matrix(os.path.join(ROOT, "matrices", "nasa4704", "nasa4704.mtx"))
for opts in ({}, {"panel_v2": 0}, {"fuse_b": 0}, {"fuse_b": 0, "panel_v2": 0}):
    e = api.Engine(0)
    for k, val in opts.items():
        e.set_option(k, val)
    e.set_matrix_csr(M, K, rp, ci, v)
    out = bench._measure(api, torch, e, M, K, 16, nnz, dev, st, 2000)
    Bh, Ch = api.init_dense_B(K, 16), api.init_dense_C(M, 16)
    e.spmm(16, 0.85, Bh, -2.06, Ch.copy(), rp_time=10)
    ns = e.spmm(16, 0.85, Bh, -2.06, Ch, rp_time=2000)
    print("nasa4704 N=16", opts, out["kernel"], "eager us/step", out["us_per_step"], "kernel_us", out["kernel_us"], "rp_time us/repeat", round(ns / 2000 / 1e3, 3), e.last_kernel(), flush=True)
    e.close()
for opts in ({}, {"fuse_b": 0}, {"fuse_b": 0, "tiles_per_wg": 2}, {"fuse_b": 0, "tiles_per_wg": 1}, {"fuse_b": 0, "tiles_per_wg": 4}, {"panel_v2": 0}):
    p, i, vv, nz = api.gen_fem3d_device(0, 35, 19, 7, 3, 3)
    e = api.Engine(0)
    for k, val in opts.items():
        e.set_option(k, val)
    e.set_matrix_csr_device(13965, 13965, nz, p, i, vv)
    out = bench._measure(api, torch, e, 13965, 13965, 128, nz, dev, st, 500)
    print("config3 stand-in N=128", opts, out["kernel"], "us/step", out["us_per_step"], "kernel_us", out["kernel_us"], "repack_us", out["repack_us"], flush=True)
    e.close()
    for q in (p, i, vv):
        api.device_free(0, q)
