#!/usr/bin/env python3
"""tools/reorder_exp.py <n> <dof> [N,N,...] -- graph clustering / reordered form on an n^3 x dof FEM matrix under different numberings
(natural grid order, random node order, RCM) and on an unstructured jittered mesh; matrices are built on the HOST (sextans_amd/meshgen.py).
Prints kernel / layout-pass / post-pass microseconds, plan seconds and the panel figures; every result is checked bitwise against the
natural-order kernels (row_cluster = 0) of the same matrix."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from sextans_amd import api, meshgen

n = int(sys.argv[1]); dof = int(sys.argv[2])
Ns = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "16,128").split(",")]
classes = (sys.argv[4] if len(sys.argv) > 4 else "natural,random,rcm,mesh_sweep,mesh_random").split(",")
iters = 10
st = torch.cuda.current_stream().cuda_stream


def alg_bytes(M, K, N, nnz):
    return 8 * nnz + 4 * (M + 1) + 4 * K * N + 8 * M * N


def matrices():
    t = time.time()
    rp, ci, v = api.gen_fem3d_host(n, n, n, dof, 3)
    M = n * n * n * dof
    print(f"# host fem {n}^3 x {dof}: M={M} nnz={rp[-1]} ({time.time() - t:.1f} s)", flush=True)
    if "natural" in classes:
        yield "natural", rp, ci, v, M
    if "random" in classes:
        t = time.time(); out = meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M // dof, dof, 1)); print(f"# random perm {time.time() - t:.1f} s", flush=True)
        yield "random", *out, M
    if "rcm" in classes:
        t = time.time(); out = meshgen.permute_symmetric(rp, ci, v, M, meshgen.rcm_node_permutation(rp, ci, M, dof)); print(f"# rcm {time.time() - t:.1f} s", flush=True)
        yield "rcm", *out, M
    del rp, ci, v
    for num in ("sweep", "random"):
        if "mesh_" + num in classes:
            t = time.time(); out = meshgen.jittered_mesh3d(n, n, n, 5, numbering=num, dof=dof); print(f"# mesh {num} {time.time() - t:.1f} s", flush=True)
            yield "mesh_" + num, *out


for name, rp, ci, v, M in matrices():
    nnz = int(rp[-1])
    for N in Ns:
        B = torch.empty(M * N, device="cuda"); Cin = torch.empty(M * N, device="cuda")
        api.gen_uniform_device(0, B.data_ptr(), M * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
        ref = None
        for rc in (0, -1, 2):
            e = api.Engine(0)
            e.set_option("row_cluster", rc)
            e.set_matrix_csr(M, M, rp, ci, v)
            Cout = torch.zeros(M * N, device="cuda")
            f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), M, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
            for _ in range(3): f()
            torch.cuda.synchronize()
            e.set_option("profile", 1); e.profile_reset()
            t0 = time.time()
            for _ in range(iters): f()
            torch.cuda.synchronize()
            wall = (time.time() - t0) / iters
            k_ns, cnt, r_ns = e.profile_read(); p_ns, _ = e.profile_read_post()
            if ref is None: ref = Cout.clone()
            same = bool(torch.equal(ref.view(torch.int32), Cout.view(torch.int32)))
            by = alg_bytes(M, M, N, nnz)
            step = (k_ns + r_ns + p_ns) * 1e-9
            print(f"{name:12s} N={N:3d} row_cluster={rc:2d} state={int(e.get_stat('row_cluster')):2d} {e.last_kernel():28s} kernel {k_ns / 1e3:8.1f} us  pre {r_ns / 1e3:7.1f}  post {p_ns / 1e3:7.1f}"
                  f"  wall/step {wall * 1e6:8.1f}  frac(kernel) {by / (k_ns * 1e-9) / 8e12:.3f} frac(step) {by / step / 8e12:.3f}  plan {e.get_stat('plan_build_s'):.3f} s"
                  f"  panel rows nat {e.get_stat('panel_rows_natural'):.0f} clu {e.get_stat('panel_rows_clustered'):.0f} blocks nat {e.get_stat('panel_blocks'):.0f} clu {e.get_stat('panel_blocks_clustered'):.0f}"
                  f"  shared {e.get_stat('cluster_shared_fraction'):.2f} dev {e.get_stat('device_bytes') / 1e9:.2f} GB  bits_equal={same}", flush=True)
            e.close()
