#!/usr/bin/env python3
"""tools/rank_slabs.py -- what every rank of a row-partitioned SpMM does on its own, measured ONE RANK AT A TIME ON ONE GPU.
NOT a scaling curve: the pool gives this build single-GPU boxes, so the slabs of world = 2 / 4 / 8 run one after the other on the
same device; no collective runs, nothing overlaps.  What it does give: the per-rank kernel and repack times that the N-GPU
compute phase is made of (max over ranks = the compute-only step of that world size), for config 4 and for the 4M-row FEM matrix,
each rank holding exactly the matrix it would hold (rows [r0, r1) generated in HBM, B replicated, a packed slab of C written) --
the reference's counterpart is the row % 64 sharding over PEs with B broadcast (sparse_helper.h:370, sextans.cpp:916-927).
Writes one JSON document to stdout."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api, dist as sxd

N = 16
CHUNKS = int(sys.argv[sys.argv.index("--chunks") + 1]) if "--chunks" in sys.argv else 1   # > 1: the chunk pipeline of sextans_dist_spmm without collectives (comm = NULL)
RM = "--rm" in sys.argv                                                                     # row-major operands: what a rank of sextans_dist_spmm_rm runs (no repack, no staging)
ONLY = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else ""               # substring of the matrix names to run
st = torch.cuda.current_stream().cuda_stream
ALPHA, BETA = 0.85, -2.06


def measure(e, m_loc, K, nnz):
    B = torch.empty(K * N, device="cuda"); Cin = torch.empty(m_loc * N, device="cuda"); Cout = torch.empty(m_loc * N, device="cuda")
    api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), m_loc * N, 42, st)
    f = lambda: e.spmm_device2(N, ALPHA, B.data_ptr(), K, BETA, Cin.data_ptr(), m_loc, Cout.data_ptr(), m_loc, st)
    if RM:
        f = lambda: e.spmm_device_rm(N, ALPHA, B.data_ptr(), N, BETA, Cin.data_ptr(), N, Cout.data_ptr(), N, st)
    if CHUNKS > 1:   # the rank's slab in CHUNKS row chunks, staged and unpacked as in a multi-rank run, no collective (world = 1, comm = NULL)
        f = lambda: e.dist_spmm(None, 1, 0, [(0, m_loc)], N, ALPHA, B.data_ptr(), K, BETA, Cin.data_ptr(), m_loc, Cout.data_ptr(), m_loc, nchunks=CHUNKS, stream=st)
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); f(); torch.cuda.synchronize()
    for _ in range(min(300, int(0.06 / max(time.perf_counter() - t0, 1e-6)))): f()   # ~60 ms of warm-up (clock ramp after an idle phase)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 10
    e.set_option("profile", 1); e.profile_reset()
    for _ in range(10): f()
    torch.cuda.synchronize()
    k, n, r = e.profile_read(); p, _ = e.profile_read_post()
    k *= max(1, n // 10)          # (chunked: one timed launch group per chunk -> the step's total)
    e.set_option("profile", 0)
    by = 8 * nnz + 4 * (m_loc + 1) + 4 * K * N + 8 * m_loc * N
    return {"rows": m_loc, "nnz": nnz, "chunks": CHUNKS, "layout": "row-major (sextans_spmm_device_rm)" if RM else "column-major", "kernel": e.last_kernel(), "kernel_us": round(k / 1e3, 1), "repack_us": round(r / 1e3, 1), "post_us": round(p / 1e3, 1),
            "us_per_step": round(wall * 1e6, 1), "alg_bytes": by, "roofline_frac_kernel": round(by / (k * 1e-9) / 8e12, 4)}


def run(name, M, K, gen, slab_offset=False, owned=3):
    # slab_offset: the engine is told where its rows sit (option row_offset; sextans_dist_spmm does that): the graph clustering then
    # also runs on a rank's slab.  owned: how many of the arrays gen() returns are this call's to free.
    out = {"matrix": name, "M": M, "K": K, "N": N, "worlds": {}}
    for world in (1, 2, 4, 8):
        ranks = []
        for (r0, r1) in sxd.partition_rows_even(M, world):
            p, i, v, nnz = gen(r0, r1)
            e = api.Engine(0)
            if slab_offset: e.set_option("row_offset", r0)
            e.set_matrix_csr_device(r1 - r0, K, nnz, p, i, v)
            m = measure(e, r1 - r0, K, nnz)
            m["row_cluster"] = int(e.get_stat("row_cluster")); m["cluster_decline"] = int(e.get_stat("cluster_decline"))
            ranks.append(dict(rank=len(ranks), row_range=[r0, r1], **m))
            e.close()
            for q in (p, i, v)[:owned]: api.device_free(0, q)
            torch.cuda.empty_cache()
        slow = max(r["us_per_step"] for r in ranks)
        out["worlds"][str(world)] = {"ranks": ranks, "max_us_per_step": slow, "max_kernel_us": max(r["kernel_us"] for r in ranks),
                                     "allgather_bytes_received_per_rank": 4 * N * (M - M // world)}
        print(f"# {name} world {world}: slowest rank {slow:.1f} us/step, kernel {out['worlds'][str(world)]['max_kernel_us']:.1f} us", file=sys.stderr, flush=True)
    base = out["worlds"]["1"]["max_us_per_step"]
    for w in out["worlds"].values():
        w["compute_only_speedup_if_ranks_ran_in_parallel"] = round(base / w["max_us_per_step"], 2)
    return out


if "--bell" in sys.argv:
    # BASELINE config 5 (blocked-ELL bf16, 1M x 1M, 328 blocks per block row, N = 256) in block-row ranges: every rank's slab through
    # sextans_dist_spmm_bell without a communicator (packed slab into the staging buffer + the unpack pass of its own rows), one after
    # the other on this GPU.  A slab is a pointer offset into the generated arrays.
    M = K = 1_048_576; W = 328; NB = 256
    dc, dv = api.gen_bell_device(0, M, K, W, 5)
    B = torch.empty(K * NB, dtype=torch.int16, device="cuda"); Cin = torch.empty(M * NB, device="cuda")
    api.gen_uniform_bf16_device(0, B.data_ptr(), K * NB, 51, st); api.gen_uniform_device(0, Cin.data_ptr(), M * NB, 52, st)
    doc = {"what": "config 5 (blocked-ELL bf16, N = 256): per-rank block-row slabs run sequentially on ONE MI355X through sextans_dist_spmm_bell (comm = NULL); not a scaling measurement", "worlds": {}}
    for world in (1, 2, 4, 8):
        ranks = []
        for g in range(world):
            r0, r1 = g * (M // world), (g + 1) * (M // world)
            e = api.Engine(0)
            e.set_matrix_bell_device(r1 - r0, K, W, dc + 4 * (r0 // 32) * W, dv + 2048 * (r0 // 32) * W)
            Cout = torch.empty((r1 - r0) * NB, device="cuda")
            f = lambda: e.dist_spmm_bell(None, 1, 0, [(0, r1 - r0)], NB, ALPHA, B.data_ptr(), K, BETA, Cin.data_ptr() + 4 * r0, M, Cout.data_ptr(), r1 - r0, stream=st)
            f(); f(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): f()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / 3
            e.set_option("profile", 1); e.profile_reset()
            for _ in range(3): f()
            torch.cuda.synchronize()
            k, n, r = e.profile_read()
            ranks.append({"rank": g, "row_range": [r0, r1], "kernel": e.last_kernel(), "kernel_ms": round(k / 1e6, 3), "repack_b_us": round(r / 1e3, 1), "ms_per_step": round(wall * 1e3, 3)})
            e.close(); del Cout
        doc["worlds"][str(world)] = {"ranks": ranks, "max_ms_per_step": max(x["ms_per_step"] for x in ranks), "allgather_bytes_received_per_rank": 4 * NB * (M - M // world)}
        print(f"# config 5 world {world}: slowest rank {doc['worlds'][str(world)]['max_ms_per_step']:.2f} ms/step", file=sys.stderr, flush=True)
    base = doc["worlds"]["1"]["max_ms_per_step"]
    for w in doc["worlds"].values(): w["compute_only_speedup_if_ranks_ran_in_parallel"] = round(base / w["max_ms_per_step"], 2)
    print(json.dumps(doc, indent=1))
    sys.exit(0)

from sextans_amd import meshgen
_base = api.gen_fem3d_device(0, 110, 110, 110, 3, 3)
_perm = api.permute_symmetric_device(0, 3_993_000, _base[3], *_base[:3], meshgen.node_permutation(3_993_000 // 3, 3, 1))
for q in _base[:3]: api.device_free(0, q)
_slab = lambda r0, r1: api.slice_rows_device(0, r0, r1, *_perm)
def _renumbered_by_engine():
    """the randomly numbered matrix renumbered ONCE by the engine's own clustered row order (sextans_export_row_order), in HBM"""
    import numpy as np
    e = api.Engine(0)
    e.set_matrix_csr_device(3_993_000, 3_993_000, _base[3], *_perm)
    order, kind = e.export_row_order()
    e.close()
    new_of_old = np.empty(3_993_000, np.int64); new_of_old[order] = np.arange(3_993_000)
    q = api.permute_symmetric_device(0, 3_993_000, _base[3], *_perm, new_of_old)
    return (lambda r0, r1: api.slice_rows_device(0, r0, r1, *q)), kind


_all = lambda name, *a: run(name, *a) if ONLY in name else None
doc = {"what": "per-rank slabs of a row-partitioned SpMM run sequentially on ONE MI355X (tools/rank_slabs.py); not a scaling measurement"
               + ("; ROW-major operands: a rank of sextans_dist_spmm_rm computes its slab in place, nothing is repacked or staged" if RM else "")
               + (f"; every slab in {CHUNKS} row chunks through sextans_dist_spmm without collectives (its own rows staged and unpacked)" if CHUNKS > 1 else ""),
       "matrices": [_all("config4: uniform 4M x 4M, Poisson(40)", 4_000_000, 4_000_000, lambda r0, r1: api.gen_csr_device(0, 4_000_000, 4_000_000, 40.0, 4, r0, r1)),
                    _all("fem3d 110x110x110 x 3 dof (natural order)", 3_993_000, 3_993_000, lambda r0, r1: api.gen_fem3d_device(0, 110, 110, 110, 3, 3, r0, r1)),
                    _all("fem3d 110x110x110 x 3 dof, random node order, slabs without their position (row-similarity graph since round 5)", 3_993_000, 3_993_000, _slab, False, 1),
                    _all("fem3d 110x110x110 x 3 dof, random node order, slabs that know their row offset (graph clustering per slab)", 3_993_000, 3_993_000, _slab, True, 1)]}
if ONLY in "renumbered by the engine":
    _slab2, _kind = _renumbered_by_engine()
    doc["matrices"].append(run(f"fem3d 110x110x110 x 3 dof, random node order, renumbered by the engine's clustered row order first (sextans_export_row_order, kind {_kind}), then contiguous row ranges", 3_993_000, 3_993_000, _slab2, False, 1))
doc["matrices"] = [m for m in doc["matrices"] if m]
print(json.dumps(doc, indent=1))
