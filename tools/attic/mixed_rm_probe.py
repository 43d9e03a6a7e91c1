#!/usr/bin/env python3
"""tools/mixed_rm_probe.py -- a matrix whose plan is MIXED (FEM rows = dictionary blocks, uniformly random rows = direct blocks): which
kernels the column-major and the row-major entry points run and what a step costs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sextans_amd import api
import bench
nx = 70
frp, fci, fv = api.gen_fem3d_host(nx, nx, nx, 3, 5)
Mf = nx * nx * nx * 3
import os
for frac in tuple(float(x) for x in os.environ.get('MIXED_SHARES', '0.1,0.3,0.6').split(',')):
    Mu = int(Mf * frac)
    urp, uci, uv = api.gen_csr_host(Mu, Mf, 40.0, 4, 0, Mu)
    rp = np.concatenate([frp, frp[-1] + urp[1:]]).astype(np.int32); ci = np.concatenate([fci, uci]); v = np.concatenate([fv, uv])
    M, K, nnz = Mf + Mu, Mf, int(rp[-1])
    dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
    for layout in ("cm", "rm"):
        e = api.Engine(0)
        for kv in sys.argv[1:]:
            k, val = kv.split("="); e.set_option(k, int(val))
        e.set_matrix_csr(M, K, rp, ci, v)
        r = bench._measure(api, torch, e, M, K, 16, nnz, dev, st, 20, layout)
        print(f"random-row share {frac} layout {layout}: kernel {r['kernel']} step {r['us_per_step']} us kernel {r['kernel_us']} repack {r['repack_us']} frac_step {r['roofline_frac_step']} panel_fraction {e.get_stat('panel_fraction'):.3f}", flush=True)
        e.close()
