#!/usr/bin/env python3
"""tools/reorder_big.py [n=110] [dof=3] [N list] [classes] -- the 4M-row FEM matrix (n^3 nodes x dof, generated in HBM) under node
renumberings applied ON THE DEVICE (sextans_csr_permute_symmetric_device): natural grid order, random node order, reverse
Cuthill-McKee (scipy on the 1-dof node graph), plus unstructured jittered meshes built on the host.  For every (class, N):
row_cluster = 0 (natural-order forms) against -1 (automatic), kernel / pre / post microseconds from HIP events, plan seconds, panel
figures, and a bitwise comparison of the two results.  One JSON record per line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from sextans_amd import api, meshgen

n = int(sys.argv[1]) if len(sys.argv) > 1 else 110
dof = int(sys.argv[2]) if len(sys.argv) > 2 else 3
Ns = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "16,128").split(",")]
classes = (sys.argv[4] if len(sys.argv) > 4 else "natural,random,rcm,mesh_sweep,mesh_random").split(",")
modes = [int(x) for x in (sys.argv[5] if len(sys.argv) > 5 else "0,-1").split(",")]
iters = 10
st = torch.cuda.current_stream().cuda_stream


def alg_bytes(M, K, N, nnz):
    return 8 * nnz + 4 * (M + 1) + 4 * K * N + 8 * M * N


def device_classes():
    M = n * n * n * dof
    base = api.gen_fem3d_device(0, n, n, n, dof, 3)
    nnz = base[3]
    if "natural" in classes:
        yield "fem_natural", M, nnz, base[:3], None
    if "random" in classes:
        t = time.time()
        p = api.permute_symmetric_device(0, M, nnz, *base[:3], meshgen.node_permutation(M // dof, dof, 1))
        print(f"# random node order applied on the device: {time.time() - t:.1f} s", flush=True)
        yield "fem_random_nodes", M, nnz, p, p
    if "rcm" in classes:
        t = time.time()
        rp1, ci1, v1 = api.gen_fem3d_host(n, n, n, 1, 3)
        perm = meshgen.rcm_node_permutation(rp1, ci1, n * n * n, 1)
        del rp1, ci1, v1
        p = api.permute_symmetric_device(0, M, nnz, *base[:3], meshgen.expand_dof(perm, dof))
        print(f"# RCM (scipy on the node graph) + device permutation: {time.time() - t:.1f} s", flush=True)
        yield "fem_rcm", M, nnz, p, p
    for q in base[:3]:
        api.device_free(0, q)


def host_classes():
    for num in ("sweep", "random"):
        if "mesh_" + num in classes:
            t = time.time()
            rp, ci, v, M = meshgen.jittered_mesh3d(n, n, n, 5, numbering=num, dof=dof)
            print(f"# jittered mesh {n}^3 x {dof} dof, numbering {num}: M={M} nnz={rp[-1]} ({time.time() - t:.1f} s on the host)", flush=True)
            yield "mesh_" + num, M, int(rp[-1]), (rp, ci, v)


def run(name, M, nnz, setter):
    for N in Ns:
        B = torch.empty(M * N, device="cuda"); Cin = torch.empty(M * N, device="cuda")
        api.gen_uniform_device(0, B.data_ptr(), M * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
        ref = None
        for rc in modes:
            e = api.Engine(0)
            e.set_option("row_cluster", rc)
            setter(e)
            Cout = torch.zeros(M * N, device="cuda")
            f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), M, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
            for _ in range(3): f()
            torch.cuda.synchronize()
            t1 = time.time(); f(); torch.cuda.synchronize()
            for _ in range(min(300, int(0.06 / max(time.time() - t1, 1e-6)))): f()   # ~60 ms of warm-up (clock ramp after an idle phase)
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
            rec = dict(matrix=name, M=M, nnz=nnz, N=N, row_cluster=rc, state=int(e.get_stat("row_cluster")), decline=int(e.get_stat("cluster_decline")),
                       kernel=e.last_kernel(), kernel_us=round(k_ns / 1e3, 1), pre_us=round(r_ns / 1e3, 1), post_us=round(p_ns / 1e3, 1),
                       wall_us=round(wall * 1e6, 1), frac_kernel=round(by / (k_ns * 1e-9) / 8e12, 4), frac_step=round(by / step / 8e12, 4),
                       plan_build_s=round(e.get_stat("plan_build_s"), 3), panel_rows_natural=int(e.get_stat("panel_rows_natural")),
                       panel_rows_clustered=int(e.get_stat("panel_rows_clustered")), panel_blocks=int(e.get_stat("panel_blocks")),
                       panel_blocks_clustered=int(e.get_stat("panel_blocks_clustered")), shared=round(e.get_stat("cluster_shared_fraction"), 3),
                       device_gb=round(e.get_stat("device_bytes") / 1e9, 2), bits_equal_to_first_mode=same)
            print(json.dumps(rec), flush=True)
            e.close()
            del Cout
        del B, Cin, ref
        torch.cuda.empty_cache()


for name, M, nnz, ptrs, owned in device_classes():
    run(name, M, nnz, lambda e: e.set_matrix_csr_device(M, M, nnz, *ptrs))
    if owned:
        for q in owned:
            api.device_free(0, q)
for name, M, nnz, host in host_classes():
    run(name, M, nnz, lambda e: e.set_matrix_csr(M, M, *host))
