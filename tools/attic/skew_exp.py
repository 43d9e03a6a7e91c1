#!/usr/bin/env python3
"""tools/skew_exp.py [key=value ...] -- 1M-row power-law matrix vs the same-nnz uniform matrix (N = 16), default engine
options unless overridden; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sextans_amd import api  # noqa: E402

opts = dict(kv.split("=") for kv in sys.argv[1:] if "=" in kv)
M = K = int(opts.pop("rows", 1_000_000))
N = 16
st = torch.cuda.current_stream().cuda_stream
B = torch.empty(K * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st)
api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)


def time_it(name, ptrs, nnz):
    with api.Engine(0) as e:
        for k, v in opts.items():
            e.set_option(k, int(v))
        e.set_matrix_csr_device(M, K, nnz, *ptrs)
        f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{name}: {nnz} nnz {dt * 1e3:.3f} ms  {e.last_kernel()}  piece-path rows {int(e.get_stat('piece_path_rows'))} "
              f"(split {int(e.get_stat('reassociated_rows'))}) L0={int(e.get_stat('bucket_threshold'))} T={int(e.get_stat('split_threshold'))} "
              f"plan_build_s {e.get_stat('plan_build_s'):.2f}", flush=True)


pl = api.gen_powerlaw_device(0, M, K, 6, 120, 400_000, 7)
time_it("power-law", pl[:3], pl[3])
for q in pl[:3]:
    api.device_free(0, q)
un = api.gen_csr_device(0, M, K, pl[3] / M, 7)
time_it("uniform  ", un[:3], un[3])
