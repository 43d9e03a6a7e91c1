#!/usr/bin/env python3
"""tools/sweep.py -- kernel-variant sweep on the GPU box (per-kernel HIP-event times).
Usage: python tools/sweep.py [workload ...]   workloads: uniform banded2k banded20k c3 nasa"""
import itertools
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sextans_amd import api  # noqa: E402

ALPHA, BETA = 0.85, -2.06


def alg_bytes(M, K, N, nnz):
    return 8 * nnz + 4 * (M + 1) + 4 * K * N + 8 * M * N


def workload(name):
    if name == "nasa":
        rp, ci, v, M, K, nnz = api.read_suitsparse_matrix(os.path.join(ROOT, "matrices/nasa4704/nasa4704.mtx"))
        return dict(M=M, K=K, N=16, host=(rp, ci, v), nnz=nnz)
    if name == "c3":
        return dict(M=13965, K=13965, N=128, fem=(35, 19, 7, 3, 2))
    if name == "fem":
        return dict(M=110 * 110 * 110 * 3, K=110 * 110 * 110 * 3, N=16, fem=(110, 110, 110, 3, 3))
    if name.startswith("femN"):   # the FEM class at another N: femN32, femN128
        return dict(M=110 * 110 * 110 * 3, K=110 * 110 * 110 * 3, N=int(name[4:]), fem=(110, 110, 110, 3, 3))
    if name == "fem1":
        return dict(M=160 ** 3, K=160 ** 3, N=16, fem=(160, 160, 160, 1, 3))
    if name == "powerlaw":
        rs = np.random.RandomState(17)
        M = K = 1_000_000
        lens = rs.poisson(20, M)
        hubs = rs.choice(M, 16, replace=False)
        lens[hubs] = 400_000
        rp = np.zeros(M + 1, np.int32); rp[1:] = np.cumsum(lens)
        ci = (rs.randint(0, K, rp[-1])).astype(np.int32)          # columns need not be sorted/distinct for timing
        v = rs.uniform(-1, 1, rp[-1]).astype(np.float32)
        return dict(M=M, K=K, N=16, host=(rp, ci, v), nnz=int(rp[-1]))
    if name == "uniform":
        return dict(M=4_000_000, K=4_000_000, N=16, gen=(40.0, 0, 4))
    if name.startswith("banded"):
        bw = int(name[6:].replace("k", "000"))
        return dict(M=4_000_000, K=4_000_000, N=16, gen=(40.0, bw, 4))
    raise SystemExit("unknown workload " + name)


def main():
    names = sys.argv[1:] or ["fem", "fem1", "banded2k", "c3", "nasa", "uniform"]
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    for name in names:
        w = workload(name)
        M, K, N = w["M"], w["K"], w["N"]
        e = api.Engine(0)
        ptrs = None
        if "host" in w:
            e.set_matrix_csr(M, K, *w["host"])
            nnz = w["nnz"]
        elif "fem" in w:
            ptrs = api.gen_fem3d_device(0, *w["fem"])
            nnz = ptrs[3]
            e.set_matrix_csr_device(M, K, nnz, *ptrs[:3])
        else:
            mean, bw, seed = w["gen"]
            ptrs = api.gen_csr_device(0, M, K, mean, seed, bandwidth=bw)
            nnz = ptrs[3]
            e.set_matrix_csr_device(M, K, nnz, *ptrs[:3])
        B = torch.empty(K * N, device=dev); Cin = torch.empty(M * N, device=dev); Cout = torch.empty(M * N, device=dev)
        api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
        by = alg_bytes(M, K, N, nnz)
        print(f"== {name}: M={M} K={K} N={N} nnz={nnz} alg_bytes={by/1e6:.1f} MB  (100% of 8 TB/s = {by/8e12*1e6:.1f} us)")
        ref = None
        for kernel, lpr, stage, xcd, exact in itertools.product((1, 2), (4, 8, 2), (1,), (1, 0), (1,)):
            if 4 * lpr > N or (lpr == 2 and N > 16):
                continue
            for k, val in dict(kernel=kernel, lanes_per_row=lpr, stage_a=stage, xcd_remap=xcd, exact=exact, profile=0).items():
                e.set_option(k, val)
            f = lambda: e.spmm_device(N, ALPHA, B.data_ptr(), K, BETA, Cin.data_ptr(), Cout.data_ptr(), M, st)
            for _ in range(3):
                f()
            e.set_option("profile", 1); e.profile_reset()
            reps = 200 if M < 100000 else 10
            for _ in range(reps):
                f()
            torch.cuda.synchronize()
            k_ns, n, r_ns = e.profile_read()
            e.set_option("profile", 0); e.profile_reset()
            chk = float(Cout.double().sum().item())
            if exact and ref is None:
                ref = Cout.clone()
            same = bool(torch.equal(ref, Cout)) if exact else None
            print(f"  kernel={kernel} lpr={lpr} stage={stage} xcd={xcd} exact={exact}: kernel {k_ns/1e3:9.2f} us  repack {r_ns/1e3:7.2f} us  "
                  f"alg {by/(k_ns*1e-9)/1e9:8.1f} GB/s ({by/(k_ns*1e-9)/8e12*100:5.1f}% of 8TB/s)  bitsame={same} sum={chk:.6g}")
        e.close()
        if ptrs:
            for q in ptrs[:3]:
                api.device_free(0, q)


if __name__ == "__main__":
    main()
