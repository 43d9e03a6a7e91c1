#!/usr/bin/env python3
"""tools/window_exp.py [rows] [n] -- K-windowed kernel (kernel=3) against the gather kernel (kernel=1) on the
config-4 generator (uniform columns): time per launch (HIP events), plan build seconds, padding, bitwise
agreement.  Sweeps rows-per-wavefront, window width and ring depth."""
import itertools
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sextans_amd import api  # noqa: E402

M = K = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
variants = sys.argv[3] if len(sys.argv) > 3 else "full"
if len(sys.argv) > 4:
    K = int(sys.argv[4])                # K < M: probe with a B panel that fits the L2s (every gather hits)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
p, i, v, nnz = api.gen_csr_device(0, M, K, 40.0, 4)
B = torch.empty(K * N, device=dev); Cin = torch.empty(M * N, device=dev)
api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
by = 8 * nnz + 4 * (M + 1) + 4 * K * N + 8 * M * N
e = api.Engine(0)
e.set_matrix_csr_device(M, K, nnz, p, i, v)


def timed(reps=10):
    out = torch.empty(M * N, device=dev)
    f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), out.data_ptr(), M, st)
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e.set_option("profile", 1); e.profile_reset()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    k_ns, n, r_ns = e.profile_read()
    e.set_option("profile", 0); e.profile_reset()
    return out, k_ns, r_ns


e.set_option("kernel", 1)
ref, k_ns, r_ns = timed()
print(f"M={M} K={K} N={N} nnz={nnz} alg={by/1e9:.3f} GB")
print(f"gather  : {e.last_kernel():18s} kernel {k_ns/1e3:9.1f} us repack {r_ns/1e3:6.1f} us  frac {by/(k_ns*1e-9)/8e12:.4f}", flush=True)
e.set_option("kernel", 3)
if variants == "full":
    combos = [(319, 65536, 8), (319, 65536, 4), (319, 32768, 8), (319, 131072, 8), (319, 16384, 8), (255, 65536, 8),
              (159, 65536, 8), (319, 4_000_000 if K <= 4_000_000 else 8_000_000, 8)]
else:
    combos = [(319, 65536, 8), (319, 65536, 4)]
for rows, cols, unroll in combos:
    e.set_option("window_rows", rows); e.set_option("window_cols", cols); e.set_option("window_unroll", unroll)
    t0 = time.perf_counter()
    out, k_ns, r_ns = timed()
    wall = time.perf_counter() - t0
    same = bool(torch.equal(out, ref))
    print(f"window  : rows={rows:3d} cols={cols:7d} unroll={unroll} {e.last_kernel():16s} kernel {k_ns/1e3:9.1f} us repack {r_ns/1e3:6.1f} us "
          f"frac {by/(k_ns*1e-9)/8e12:.4f} padded/nnz {e.get_stat('window_padded_entries')/nnz:.4f} "
          f"plan_build_s(total) {e.get_stat('plan_build_s'):.1f} wall {wall:.1f}s bitsame={same}", flush=True)
e.set_option("kernel", 0)
out, k_ns, r_ns = timed()
print(f"auto    : {e.last_kernel():18s} kernel {k_ns/1e3:9.1f} us repack {r_ns/1e3:6.1f} us  frac {by/(k_ns*1e-9)/8e12:.4f} bitsame={bool(torch.equal(out, ref))}")
