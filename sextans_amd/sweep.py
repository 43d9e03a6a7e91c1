"""SuiteSparse sweep harness: the paper-style evaluation loop over many .mtx
files and N values -- what a user of the reference does by calling `sextans <A.mtx> <N>` in a shell
loop (README.md:18,31) -- with one JSON record per (matrix, N).

    python -m sextans_amd.sweep matrices/nasa4704/nasa4704.mtx more/*.mtx --n 8,16,32,64,128 --rp 50

Each record: matrix name, M, K, nnz, N, kernel chosen by the dispatcher, device ms per SpMM
(rp_time repeats, sextans-host.cpp:252 convention), GFLOP/s with the reference's formula
2*N*(nnz+M) (sextans-host.cpp:255-260), algorithmic GB/s and fraction of the 8 TB/s HBM roofline
(8*nnz + 4*(M+1) + 4*K*N + 8*M*N bytes), and -- with --check -- the reference's pass criterion
(mismatch % < 2, sextans-host.cpp:272-282) against the host golden, like the CLI's self check.
GPU only: there is no CPU path for the product computation.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

from . import api

HBM_PEAK_GBS = 8000.0


def sweep(paths, n_values, rp_time=20, alpha=0.85, beta=-2.06, check=False, device=0, options=None, out=sys.stdout,
          cache=False, inspect=None):
    """inspect(record, M, K, N, row_ptr, col_idx, val, B, C_in, C_out): optional hook called with every result
    (the tests compare C_out with the oracle there)."""
    records = []
    with api.Engine(device) as eng:
        for k, v in (options or {}).items():
            eng.set_option(k, v)
        for path in paths:
            name = os.path.splitext(os.path.basename(path))[0]
            try:
                t0 = time.perf_counter()
                rp, ci, va, M, K, nnz = api.read_suitsparse_matrix(path, cache=True if cache else None)
                load_s = time.perf_counter() - t0
            except api.SextansError as e:
                rec = {"matrix": name, "error": str(e)}
                records.append(rec); print(json.dumps(rec), file=out, flush=True)
                continue
            if M == 0 or K == 0:
                continue
            eng.set_matrix_csr(M, K, rp, ci, va)
            for n in n_values:
                N = api.round_up_n(n)
                B = api.init_dense_B(K, N)
                C0 = api.init_dense_C(M, N)
                C = C0.copy()
                ns = eng.spmm(N, alpha, B, beta, C, rp_time=rp_time)
                sec = ns * 1e-9 / max(rp_time, 1)
                by = 8 * nnz + 4 * (M + 1) + 4 * K * N + 8 * M * N
                rec = {"matrix": name, "M": M, "K": K, "nnz": nnz, "N": N, "kernel": eng.last_kernel(),
                       "load_s": round(load_s, 4), "ms": round(sec * 1e3, 6), "gflops": round(api.gflops(M, N, nnz, sec), 2),
                       "alg_gbs": round(by / sec / 1e9, 2),
                       "roofline_frac": round(by / sec / 1e9 / HBM_PEAK_GBS, 5)}
                if check:
                    gold = C0.copy()
                    rc = api.lib().sextans_selfcheck_golden(M, N, K, alpha, rp, api._buf(ci, np.int32),
                                                            api._buf(va, np.float32), B, beta, gold)
                    mism, pct = api.verify(M, N, gold, C)
                    rec.update(mismatch=int(mism), mismatch_pct=round(float(pct), 4), passed=bool(rc == 0 and pct < 2.0),
                               bit_identical=bool(np.array_equal(gold.view(np.uint32), C.view(np.uint32))))
                if inspect is not None:
                    inspect(rec, M, K, N, rp, ci, va, B, C0, C)
                records.append(rec)
                print(json.dumps(rec), file=out, flush=True)
    return records


SYNTH_HELP = """synthetic classes (generated in HBM by csrc/synth.hip, seeded, same bits on host and device):
  synth:uniform:M:mean[:K]          Poisson(mean) non-zeros per row, uniform columns (BASELINE config 4 = synth:uniform:4000000:40)
  synth:banded:M:mean:bw            the same inside the band |row - col| <= bw (locality)
  synth:fem3d:nx:ny:nz:dof          27-point node stencil with dof unknowns per node (SuiteSparse FEM class)
  synth:powerlaw:M:xmin:tail_x100:maxlen   P(len >= x) = (xmin/x)^(tail/100): hub rows over a mass of short rows
  synth:stencil2d:nx:ny:points:dof  5- or 9-point 2-D grid stencil, dof unknowns per node
  synth:kkt:n:arrow                 KKT / arrow blocks: n variables (pentadiagonal H), n/2 constraints, `arrow` dense borders
numberings without a grid (sextans_amd/meshgen.py; the permutation is computed on the host and applied in HBM):
  synth:femperm:nx:ny:nz:dof:random the fem3d matrix under a seeded random renumbering of its nodes
  synth:femperm:nx:ny:nz:dof:rcm    the same under reverse Cuthill-McKee of the node graph (scipy)
  synth:mesh3d:n:dof:sweep|random|rcm   unstructured jittered-point mesh (n^3 points, ~14 neighbours each), built on the host
holdout class with the local structure of a real SuiteSparse matrix (sextans_amd/holdout.py; round 5):
  synth:kron:n[:sym|rect|unsym[:natural|random|rcm]]   kron(T_n, nasa4704): n copies of the nasa4704 pattern coupled tridiagonally
                                                       (n = 850: 4.0 M rows, 267 M non-zeros); rect = every third column dropped,
                                                       unsym = 30 % of the strictly lower entries dropped"""


def _synth(spec, device):
    f = spec.split(":")
    kind = f[1]
    if kind == "femperm":
        from . import meshgen
        nx, ny, nz, dof = (int(x) for x in f[2:6])
        M = K = nx * ny * nz * dof
        p, i, v, nnz = api.gen_fem3d_device(device, nx, ny, nz, dof, 3)
        if f[6] == "random":
            perm = meshgen.node_permutation(M // dof, dof, 1)
        elif f[6] == "rcm":
            rp1, ci1, _ = api.gen_fem3d_host(nx, ny, nz, 1, 3)
            perm = meshgen.expand_dof(meshgen.rcm_node_permutation(rp1, ci1, nx * ny * nz, 1), dof)
        else:
            raise ValueError("femperm numbering: random | rcm")
        q = api.permute_symmetric_device(device, M, nnz, p, i, v, perm)
        for old in (p, i, v):
            api.device_free(device, old)
        return (M, K) + q + (nnz,)
    if kind == "kron":
        from . import holdout
        return holdout.kron_device(device, int(f[2]), f[3] if len(f) > 3 else "", f[4] if len(f) > 4 else "natural")
    if kind == "mesh3d":
        from . import meshgen
        n, dof = int(f[2]), int(f[3])
        rp, ci, v, M = meshgen.jittered_mesh3d(n, n, n, 5, numbering=f[4], dof=dof)
        return (M, M) + api.upload_csr(device, rp, ci, v) + (int(rp[-1]),)
    a = [int(x) for x in f[2:] if "." not in x]
    if kind == "uniform":
        M, K = a[0], (a[2] if len(a) > 2 else a[0])
        return (M, K) + api.gen_csr_device(device, M, K, float(f[3]), 4)
    if kind == "banded":
        M = K = a[0]
        return (M, K) + api.gen_csr_device(device, M, K, float(f[3]), 4, bandwidth=a[2])
    if kind == "fem3d":
        M = K = a[0] * a[1] * a[2] * a[3]
        return (M, K) + api.gen_fem3d_device(device, a[0], a[1], a[2], a[3], 3)
    if kind == "powerlaw":
        M = K = a[0]
        return (M, K) + api.gen_powerlaw_device(device, M, K, a[1], a[2], a[3], 7)
    if kind == "stencil2d":
        M = K = a[0] * a[1] * a[3]
        return (M, K) + api.gen_stencil2d_device(device, a[0], a[1], a[2], a[3], 3)
    if kind == "kkt":
        M = K = api.kkt_rows(a[0], a[1])
        return (M, K) + api.gen_kkt_device(device, a[0], a[1], 3)
    raise ValueError("unknown synthetic class: " + spec)


def sweep_synthetic(specs, n_values, steps=20, alpha=0.85, beta=-2.06, device=0, options=None, out=sys.stdout, layout="cm"):
    """The same record per (matrix, N) for matrices generated directly in HBM: device-resident B/C (seeded U(-1,1)),
    steady-state step time (repack + kernels) over `steps` back-to-back steps.  layout "rm": the operands read as ROW-major
    K x N / M x N through sextans_spmm_device_rm (records carry "layout": "rm")."""
    import torch
    records = []
    dev = torch.device("cuda", device)
    st = torch.cuda.current_stream(dev).cuda_stream
    for spec in specs:
        M, K, p, i, v, nnz = _synth(spec, device)
        with api.Engine(device) as eng:
            for k, val in (options or {}).items():
                eng.set_option(k, val)
            eng.set_matrix_csr_device(M, K, nnz, p, i, v)
            for n in n_values:
                N = api.round_up_n(n)
                B = torch.empty(K * N, device=dev); Cin = torch.empty(M * N, device=dev); Cout = torch.empty(M * N, device=dev)
                api.gen_uniform_device(device, B.data_ptr(), K * N, 41, st)
                api.gen_uniform_device(device, Cin.data_ptr(), M * N, 42, st)
                f = lambda: eng.spmm_device(N, alpha, B.data_ptr(), K, beta, Cin.data_ptr(), Cout.data_ptr(), M, st)
                if layout == "rm":
                    f = lambda: eng.spmm_device_rm(N, alpha, B.data_ptr(), N, beta, Cin.data_ptr(), N, Cout.data_ptr(), N, st)
                for _ in range(3):
                    f()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                f()
                torch.cuda.synchronize(dev)
                for _ in range(min(300, int(0.06 / max(time.perf_counter() - t0, 1e-6)))):   # ~60 ms of warm-up: the first launches after an idle phase run slower
                    f()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(steps):
                    f()
                torch.cuda.synchronize(dev)
                sec = (time.perf_counter() - t0) / steps
                by = 8 * nnz + 4 * (M + 1) + 4 * K * N + 8 * M * N
                rec = {"matrix": spec, "M": M, "K": K, "nnz": nnz, "N": N, "kernel": eng.last_kernel(),
                       "ms": round(sec * 1e3, 5), "gflops": round(api.gflops(M, N, nnz, sec), 1),
                       "alg_gbs": round(by / sec / 1e9, 1), "roofline_frac": round(by / sec / 1e9 / HBM_PEAK_GBS, 4),
                       "piece_path_rows": int(eng.get_stat("piece_path_rows")),
                       "reassociated_rows": int(eng.get_stat("reassociated_rows")),
                       "dense_tile_fraction": round(eng.get_stat("dense_tile_fraction"), 4),
                       "plan_build_s": round(eng.get_stat("plan_build_s"), 3),
                       # row order of the LDS-panel plan: 1 grid bricks, 2 graph clustering (reordered form), -1 natural
                       "row_cluster": int(eng.get_stat("row_cluster")),
                       "panel_rows_natural": int(eng.get_stat("panel_rows_natural")),
                       "panel_rows_clustered": int(eng.get_stat("panel_rows_clustered")),
                       "cluster_decline": int(eng.get_stat("cluster_decline")), "layout": layout}
                records.append(rec)
                print(json.dumps(rec), file=out, flush=True)
                del B, Cin, Cout
        for q in (p, i, v):
            api.device_free(device, q)
        torch.cuda.empty_cache()
    return records


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter, epilog=SYNTH_HELP)
    ap.add_argument("paths", nargs="+", help=".mtx files or globs, and/or synth:<class>:... specs (see below)")
    ap.add_argument("--n", default="8,16,32,64,128", help="comma-separated N values (rounded up to 8)")
    ap.add_argument("--rp", type=int, default=20, help="rp_time repeats per measurement")
    ap.add_argument("--alpha", type=float, default=0.85)
    ap.add_argument("--beta", type=float, default=-2.06)
    ap.add_argument("--check", action="store_true", help="compare with the host golden (reference criterion)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--cache", action="store_true",
                    help="read each matrix through its binary container (<file>.csr.sxbin), writing it on first use")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value")
    ap.add_argument("--rm", action="store_true", help="synthetic classes: row-major operands (sextans_spmm_device_rm)")
    a = ap.parse_args(argv)
    paths, synth = [], []
    for p in a.paths:
        if p.startswith("synth:"):
            synth.append(p)
        else:
            paths.extend(sorted(glob.glob(p)) or [p])
    opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.opt}
    ns = [int(x) for x in a.n.split(",")]
    if paths:
        sweep(paths, ns, a.rp, a.alpha, a.beta, a.check, a.device, opts, cache=a.cache)
    if synth:
        sweep_synthetic(synth, ns, max(a.rp, 1), a.alpha, a.beta, a.device, opts, layout="rm" if a.rm else "cm")


if __name__ == "__main__":
    main()
