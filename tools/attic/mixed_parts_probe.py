import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from sextans_amd import api
import bench
nx = 70
frp, fci, fv = api.gen_fem3d_host(nx, nx, nx, 3, 5)
Mf = nx * nx * nx * 3
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
e = api.Engine(0); e.set_matrix_csr(Mf, Mf, frp, fci, fv)
r = bench._measure(api, torch, e, Mf, Mf, 16, int(frp[-1]), dev, st, 20, "cm"); print("fem alone cm", r["kernel"], r["kernel_us"], r["us_per_step"]); e.close()
for frac in (0.1, 0.6):
    Mu = int(Mf * frac)
    urp, uci, uv = api.gen_csr_host(Mu, Mf, 40.0, 4, 0, Mu)
    e = api.Engine(0); e.set_matrix_csr(Mu, Mf, urp, uci, uv)
    r = bench._measure(api, torch, e, Mu, Mf, 16, int(urp[-1]), dev, st, 20, "cm"); print("random rows alone", frac, r["kernel"], r["kernel_us"], r["us_per_step"], "nnz", int(urp[-1])); e.close()
