#!/usr/bin/env python3
"""bench.py -- SpMM throughput of the MI355X engine on BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE config 4, the configuration the metric "SpMM GFLOP/s + achieved
HBM GB/s (fp32, N=16) at 1/2/4/8 MI355X" is quoted on and which fits one GPU: synthetic CSR
4,000,000 x 4,000,000, Poisson(40) non-zeros per row, U(-1,1) values (counter-based generator,
seed 4, generated directly in HBM), N = 16, fp32, alpha = 0.85, beta = -2.06, column-major B and C.
At N > 1 the SAME matrix is row-range partitioned (strong scaling), B is replicated, every rank
computes its slab of C in place and one RCCL all-gather completes C on all ranks.

A "step" is one full pass  C_out = alpha*A*B + beta*C_in  from the API layout (column-major B in
HBM -> B panel repack -> CSR row-group kernel -> column-major C_out [-> all-gather]).  FLOPs use the
reference's convention 2*N*(nnz+M) (sextans-host.cpp:255-260).  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      dominant kernel (spmm_csr_rowgroup): algorithmic bytes per launch
                (8*nnz + 4*(M+1) + 4*K*N + 8*M*N) / mean launch duration measured
                with HIP events on the launch stream, against the 8 TB/s HBM3E spec peak.
  cpu_baseline  the reference's cpu_spmm_CSR (oracle/_ref, kind "reference") or our C restatement
                (kind "port") timed single-threaded on a bounded row sample of the same workload.
  also          secondary measurements: BASELINE config 2 (nasa4704 N=16), config 3 (pcrystk02 N=128 on
                its labelled stand-in, a 35x19x7 3-dof FEM grid: 13965 rows, 968715 nnz), and a
                SuiteSparse-like 4M-row FEM matrix (the class with B-row reuse), config 5 (blocked-ELL bf16
                MFMA path, full size); compute-only time at N > 1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALPHA, BETA = 0.85, -2.06
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the copy ceiling


def alg_bytes(M, K, N, nnz, beta_nonzero=True):
    return 8 * nnz + 4 * (M + 1) + 4 * K * N + 4 * M * N * (2 if beta_nonzero else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=4_000_000, help="M = K of the synthetic matrix")
    ap.add_argument("--mean-nnz", type=float, default=40.0)
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--only", default="", help="run ONE secondary workload (its `also` key) without the headline and print its record")
    ap.add_argument("--chunks", type=int, default=4, help="N>1: row chunks for gather/compute overlap (1 = off)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample time")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value")
    ap.add_argument("--native-dist", action="store_true", help="(default since round 5; kept for old command lines)")
    ap.add_argument("--torch-dist", action="store_true",
                    help="N>1: torch.distributed collectives (PipelinedSlabGather) instead of sextans_dist_spmm, the native form "
                         "behind the C ABI (RCCL called from the library), which is the default")
    ap.add_argument("--rowmajor", action="store_true",
                    help="the same workload with ROW-major B and C (sextans_spmm_device_rm; N>1: sextans_dist_spmm_rm, slabs in place, in-place all-gather). "
                         "Not the default: BASELINE.json's configuration is column-major")
    ap.add_argument("--even-rows", action="store_true", help="N>1: equal row counts per rank instead of nnz-balanced ranges")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from sextans_amd import api, dist as sxd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves -- the same command line under
        # torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1); rank 0 of it prints the one JSON line.
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} needs {args.gpus} devices, found {have}", file=sys.stderr)
            sys.exit(2)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus}` (it spawns its ranks) "
                  f"or under torch.distributed.run with --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    api.lib()   # raises if the HIP library was not built: no fallback
    if not torch.cuda.is_available() or api.device_count() < 1:
        raise SystemExit("bench.py: no MI355X visible; the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SEXTANS_BENCH_FORCE_DIST=1 runs the multi-rank code path (process group, chunked slab, collectives,
    # unpack) on a 1-rank group: the only way to exercise it on a single-GPU box.
    multi = world > 1 or os.environ.get("SEXTANS_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.only:
        stream0 = torch.cuda.current_stream().cuda_stream
        todo = dict(secondaries(api, torch, dev, stream0, args))
        if args.only == "list":
            print(" ".join(todo))
            return
        if args.only not in todo:
            raise SystemExit("bench.py --only: unknown entry; one of " + ", ".join(todo))
        print(json.dumps({args.only: todo[args.only]()}), flush=True)
        return
    M = K = args.rows
    N = api.round_up_n(args.n)
    args.native_dist = not args.torch_dist
    if args.rowmajor and args.torch_dist:
        raise SystemExit("bench.py: --rowmajor uses the native collectives (sextans_dist_spmm_rm); drop --torch-dist")
    ranges = sxd.partition_rows_even(M, world)
    if multi and not args.even_rows:
        # north_star / SURVEY 8e: contiguous NNZ-BALANCED row ranges.  Every rank generates the row lengths of an even slice in HBM
        # (the counter-based generator yields any row range), the per-row counts are all-gathered, and the split points are found by
        # binary search in the whole matrix's row_ptr (sextans_partition_rows_by_nnz / dist.partition_rows_by_nnz).
        e0, e1 = ranges[rank]
        t_rp, t_ci, t_v, _ = api.gen_csr_device(local_rank, M, K, args.mean_nnz, 4, e0, e1)
        lens = torch.empty(e1 - e0 + 1, dtype=torch.int32, device=dev)
        api.device_copy(local_rank, lens.data_ptr(), t_rp, 4 * (e1 - e0 + 1))   # (through the library's own HIP runtime)
        for q in (t_rp, t_ci, t_v):
            api.device_free(local_rank, q)
        ranges = sxd.balanced_ranges_from_even_slices(lens, M, rank)
        del lens
    r0, r1 = ranges[rank]
    m_loc = r1 - r0

    # ---- inputs, generated in HBM
    d_rp, d_ci, d_v, nnz_loc = api.gen_csr_device(local_rank, M, K, args.mean_nnz, 4, r0, r1)
    B = torch.empty(K * N, dtype=torch.float32, device=dev)
    Cin = torch.empty(M * N, dtype=torch.float32, device=dev)
    Cout = torch.zeros(M * N, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    api.gen_uniform_device(local_rank, B.data_ptr(), K * N, 41, stream)
    api.gen_uniform_device(local_rank, Cin.data_ptr(), M * N, 42, stream)
    torch.cuda.synchronize()

    eng = api.Engine(local_rank)
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    if multi:   # hub-split threshold (only used with --opt split_rows=-1) from the whole matrix, as on one GPU
        eng.set_option("global_nnz", sxd.global_nnz(nnz_loc, dev))
    eng.set_matrix_csr_device(m_loc, K, nnz_loc, d_rp, d_ci, d_v)
    cin_ptr = Cin.data_ptr() + 4 * r0
    cout_ptr = Cout.data_ptr() + 4 * r0
    # N > 1: the rank's slab is written packed (ldc_out = rows per rank) into its slot of a staging
    # buffer, ONE RCCL all-gather moves all slabs, a local strided copy writes column-major C_out.
    # With --chunks > 1 (default 4) the slab is produced in row chunks and the all-gather of chunk i
    # overlaps the SpMM of chunk i+1 (PipelinedSlabGather).
    sg = pg = comm = None
    if multi and args.native_dist:
        ids = [api.dist_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)              # the 128-byte RCCL id travels through the torch store
        comm = api.dist_comm_init(local_rank, world, rank, ids[0])
    elif multi:
        if args.chunks > 1:
            # chunk cuts snapped to the boundaries the rank's kernel wants (row blocks of the LDS-panel plan,
            # wavefronts of the window kernel): every chunk keeps the whole-matrix kernel
            pg = sxd.PipelinedSlabGather(M, N, ranges, rank, dev, nchunks=args.chunks,
                                         align=lambda r: eng.align_row(N, r))
        else:
            sg = sxd.SlabGather(M, N, ranges, rank, dev)

    def chunk(c0, c1, out_ptr, ld_out, first):
        eng.spmm_device_rows(N, ALPHA, B.data_ptr(), K, BETA, cin_ptr + 4 * c0, M, out_ptr, ld_out, c0, c1,
                             reuse_b_panels=not first, stream=stream)

    def compute():
        if args.rowmajor:   # row-major B (K x N) and C (M x N): the rank's rows are one contiguous run of C
            eng.spmm_device_rm(N, ALPHA, B.data_ptr(), N, BETA, Cin.data_ptr() + 4 * r0 * N, N, Cout.data_ptr() + 4 * r0 * N, N, stream)
        elif not multi:
            eng.spmm_device(N, ALPHA, B.data_ptr(), K, BETA, cin_ptr, cout_ptr, M, stream)
        elif pg is not None:
            for i, ((c0, c1), S, lmax) in enumerate(zip(pg.chunks, pg.S, pg.lmax)):
                chunk(c0, c1, S[rank].data_ptr(), lmax, i == 0)
        elif comm is not None:   # compute-only leg of the native form: the rank's slab, in place, no collective
            eng.spmm_device2(N, ALPHA, B.data_ptr(), K, BETA, cin_ptr, M, cout_ptr, M, stream)
        else:
            eng.spmm_device2(N, ALPHA, B.data_ptr(), K, BETA, cin_ptr, M, sg.local_ptr(), sg.lmax, stream)

    def step():
        if not multi:
            compute()
        elif args.rowmajor:
            eng.dist_spmm_rm(comm, world, rank, ranges, N, ALPHA, B.data_ptr(), N, BETA, Cin.data_ptr(), N, Cout.data_ptr(), N, stream=stream)
        elif comm is not None:
            eng.dist_spmm(comm, world, rank, ranges, N, ALPHA, B.data_ptr(), K, BETA, Cin.data_ptr(), M, Cout.data_ptr(), M,
                          nchunks=args.chunks, stream=stream)
        elif pg is not None:
            pg.run(chunk, _force=(world == 1))
            pg.finish(Cout)
        else:
            compute()
            sg.gather(_force=(world == 1))
            sg.unpack_into(Cout)

    def barrier():
        if multi:
            dist.barrier()

    def timed(fn, iters):
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize(); barrier()
        dt = time.perf_counter() - t0
        if multi:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    if comm is not None:
        # Collective preparation OUTSIDE the timed region (round 6): non-zero counts, "row_offset", this rank's plan build, cut lists,
        # clustered-order flags and tables, workspaces -- and an error on every rank if any rank failed, instead of a hang.
        eng.dist_prepare(comm, world, rank, ranges, N, nchunks=args.chunks, form=1 if args.rowmajor else 0, stream=stream)
        setup_exchanges = eng.get_stat("dist_setup_exchanges")
    for _ in range(args.warmup):
        step()
    dt = timed(step, args.steps)
    if comm is not None:   # nothing but data collectives inside the steps
        assert eng.get_stat("dist_setup_exchanges") == setup_exchanges, "a timed step exchanged control data or synchronised with the host"
    dt_compute = timed(compute, args.steps) if multi else dt

    nnz_tot = nnz_loc
    if multi:
        t = torch.tensor([nnz_loc], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        nnz_tot = int(t.item())
    flops = 2.0 * N * (nnz_tot + M)
    sec_per_step = dt / args.steps
    value = flops / sec_per_step / 1e9

    # ---- dominant-kernel roofline: HIP events around each kernel launch on the launch stream
    eng.set_option("profile", 1)
    eng.profile_reset()
    for _ in range(10):
        compute()
    torch.cuda.synchronize()
    k_ns, k_n, rp_ns = eng.profile_read()
    eng.set_option("profile", 0)
    eng.profile_reset()
    # at N > 1 a step launches the kernel once per row chunk: k_ns is the mean per launch, so scale to
    # the whole slab (k_n launches were timed over 10 steps)
    launches_per_step = max(1, k_n // 10)
    k_ns *= launches_per_step
    bytes_launch = alg_bytes(m_loc, K, N, nnz_loc)
    achieved = bytes_launch / (k_ns * 1e-9) / 1e9
    # HBM traffic of this kernel comes from separate rocprofv3 --pmc passes of this same command
    # (tools/prof.sh -> profiles/*_traffic.json); only valid for the default single-GPU workload.
    # It is NOT measured in this run: the value and its provenance are reported side by side.
    traffic = traffic_source = None
    tpath = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_config4_traffic.json") for r in (6, 5, 4, 3)) if os.path.exists(q)), "")
    if world == 1 and args.rows == 4_000_000 and args.mean_nnz == 40.0 and N == 16 and not args.opt \
            and os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("kernel") == eng.last_kernel():
            traffic = tj.get("traffic_bytes_per_launch")
            traffic_source = ("stored, not measured in this run: profiles/" + os.path.basename(tpath) + " <- " +
                              str(tj.get("source")))
        else:   # the dispatcher picked another kernel than the one the counters were collected on: say so, loudly
            traffic_source = (f"NOT APPLICABLE: profiles/{os.path.basename(tpath)} was collected on kernel {tj.get('kernel')!r}, this run's dominant kernel is "
                              f"{eng.last_kernel()!r} -- re-run tools/prof.sh")
            print("bench.py: WARNING: " + traffic_source, file=sys.stderr)
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "kernel": eng.last_kernel(), "kernel_us": round(k_ns / 1e3, 2),
                "alg_bytes_per_launch": bytes_launch, "launches_timed": k_n,
                "kernel_launches_per_step": launches_per_step,
                "repack_us": round(rp_ns / 1e3, 2)}

    out = {
        "metric": "SpMM GFLOP/s + achieved HBM GB/s (fp32, N=16) at 1/2/4/8 MI355X",
        "value": round(value, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(sec_per_step * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"config4: synthetic CSR {M}x{K}, Poisson({args.mean_nnz:g}) nnz/row, "
                               f"U(-1,1) fp32, N={N}, alpha=0.85, beta=-2.06, {'ROW-major B/C (--rowmajor)' if args.rowmajor else 'column-major B/C'}, seed 4",
                   "M": M, "K": K, "N": N, "nnz": nnz_tot,
                   "parallelism": (f"A row-split x{world} ({'equal rows' if args.even_rows else 'nnz-balanced ranges'}), B replicated, "
                                   f"all-gather(C) by {'sextans_dist_spmm_rm (in place, RCCL from the C ABI)' if args.rowmajor else 'sextans_dist_spmm (RCCL from the C ABI)' if args.native_dist else 'torch.distributed'}")
                   if world > 1 else "1 GPU"},
        "hbm_gbs_algorithmic_step": round(alg_bytes(M, K, N, nnz_tot) / sec_per_step / 1e9, 1),
        "roofline": roofline,
    }
    out["plan_build_s"] = round(eng.get_stat("plan_build_s"), 3)   # host packing of A, outside every timed region
    also = {}
    if multi:
        also["compute_only_ms_per_step"] = round(dt_compute / args.steps * 1e3, 4)
        also["compute_only_gflops"] = round(flops / (dt_compute / args.steps) / 1e9, 2)
        also["end_to_end_ms_per_step"] = round(sec_per_step * 1e3, 4)
        # per-rank costs (rank 0's; the B repack is replicated on every rank and does not shrink with N)
        also["per_rank"] = {"rows": m_loc, "nnz": nnz_loc, "kernel": eng.last_kernel(),
                            "kernel_us_per_step": round(k_ns / 1e3, 2), "repack_us_per_step": round(rp_ns / 1e3, 2),
                            "allgather_bytes_received": 4 * N * (M - m_loc), "chunks": args.chunks,
                            "collectives": "sextans_dist_spmm_rm: in-place all-gather of row runs (RCCL from the C ABI)" if args.rowmajor
                            else "sextans_dist_spmm (RCCL from the C ABI)" if comm is not None
                            else "torch.distributed all_gather_into_tensor"}

    if rank == 0 and world > 1 and not args.no_cpu_baseline:
        # N > 1: the same CPU leg on a smaller bounded sample (the other ranks wait at the final barrier meanwhile);
        # C_out is complete on every rank after the all-gather, so rank 0 also checks the first rows bit for bit
        args.cpu_seconds = min(args.cpu_seconds, 5.0)
        out["cpu_baseline"] = cpu_baseline(api, M, K, N, args, Cout, flops_per_row=None, all_cores=False, rowmajor=args.rowmajor)
    if rank == 0 and world == 1:
        if multi:   # forced single-rank distributed run: check the gathered C against the plain path
            ref = torch.empty_like(Cout)
            if args.rowmajor:
                eng.spmm_device_rm(N, ALPHA, B.data_ptr(), N, BETA, Cin.data_ptr(), N, ref.data_ptr(), N, stream)
            else:
                eng.spmm_device(N, ALPHA, B.data_ptr(), K, BETA, cin_ptr, ref.data_ptr(), M, stream)
            torch.cuda.synchronize()
            also["dist_path_matches_plain_path"] = bool(torch.equal(ref, Cout))
            del ref
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(api, M, K, N, args, Cout, flops_per_row=None, rowmajor=args.rowmajor)
        del B, Cin, Cout
        torch.cuda.empty_cache()
        for key, fn in () if args.no_also else secondaries(api, torch, dev, stream, args):
            try:
                also[key] = fn()
            except Exception as e:   # secondary measurements only
                also[key] = {"error": str(e)}
    # The secondary measurements travel on a line of their own IN FRONT of the headline (and into gpurun_out/bench_also.json when that
    # directory exists): the LAST stdout line is the one JSON object of the contract, short enough for a log tail, with one number per
    # secondary workload (fraction of the HBM roofline per step, algorithmic bytes) so that nothing is hidden by the split.
    also_line = None
    if also:
        also_line = json.dumps({"also": also})
        out["also_step_frac"] = {k: (v.get("roofline_frac_step", v.get("roofline_frac_kernel")) if isinstance(v, dict) else v) for k, v in also.items()
                                 if not isinstance(v, dict) or "roofline_frac_step" in v or "roofline_frac_kernel" in v or "error" in v}
        out["also_line"] = "the line in front of this one: {\"also\": {...}} with every secondary record in full"

    if comm is not None:
        api.dist_comm_destroy(comm)
    eng.close()
    for q in (d_rp, d_ci, d_v):
        api.device_free(local_rank, q)
    # RCCL prints a version banner through C stdio (fully buffered when stdout is a pipe): push it out on
    # every rank BEFORE the result so that the JSON line is the last thing on stdout.
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if multi:
        dist.barrier()
    if rank == 0:
        if also_line:
            print(also_line, flush=True)
            try:
                if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
                    with open(os.path.join(ROOT, "gpurun_out", "bench_also.json"), "w") as f:
                        f.write(also_line + "\n")
            except OSError:
                pass
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()


def secondaries(api, torch, dev, stream, args):
    """The `also` entries: (key, thunk).  `python bench.py --only <key>` runs ONE of them without the headline workload (one rocprofv3
    pass per entry then gives per-workload kernel statistics: tools/evidence_r05.sh)."""
    return (("config2_nasa4704_N16", lambda: nasa_secondary(api, torch, dev, stream)),
                        ("config3_pcrystk02_surrogate_N128",
                         lambda: fem_secondary(api, torch, dev, stream, (35, 19, 7, 3), 128, 300, rp_protocol=True)),
                        ("rowmajor_config2_nasa4704_N16", lambda: nasa_secondary(api, torch, dev, stream, layout="rm")),
                        ("rowmajor_config3_pcrystk02_surrogate_N128",
                         lambda: fem_secondary(api, torch, dev, stream, (35, 19, 7, 3), 128, 300, layout="rm")),
                        ("suitesparse_like_fem_4M_N16",
                         lambda: fem_secondary(api, torch, dev, stream, (110, 110, 110, 3), 16, 100)),
                        ("suitesparse_like_fem_4M_N32",
                         lambda: fem_secondary(api, torch, dev, stream, (110, 110, 110, 3), 32, 30)),
                        ("suitesparse_like_fem_4M_N128",
                         lambda: fem_secondary(api, torch, dev, stream, (110, 110, 110, 3), 128, 10)),
                        ("suitesparse_like_fem_4M_N128_exact0_opt_in",
                         lambda: fem_secondary(api, torch, dev, stream, (110, 110, 110, 3), 128, 10, options={"exact": 0})),
                        ("fem_4M_random_node_order_N16",
                         lambda: fem_secondary(api, torch, dev, stream, (110, 110, 110, 3), 16, 40, numbering="random")),
                        ("rowmajor_fem_4M_N16",
                         lambda: fem_secondary(api, torch, dev, stream, (110, 110, 110, 3), 16, 100, layout="rm", torch_op=True)),
                        ("rowmajor_fem_4M_N128", lambda: fem_secondary(api, torch, dev, stream, (110, 110, 110, 3), 128, 10, layout="rm")),
                        ("rowmajor_fem_random_N16",
                         lambda: fem_secondary(api, torch, dev, stream, (110, 110, 110, 3), 16, 40, numbering="random", layout="rm", torch_op=True)),
                        ("renumbered_by_engine_order_fem_random_N16", lambda: renumbered_secondary(api, torch, dev, stream, "fem_random", 16, 40)),
                        ("renumbered_by_engine_order_fem_random_rowmajor_N16", lambda: renumbered_secondary(api, torch, dev, stream, "fem_random", 16, 40, "rm")),
                        ("renumbered_by_engine_order_holdout_N16", lambda: renumbered_secondary(api, torch, dev, stream, "holdout", 16, 40)),
                        ("renumbered_by_engine_order_holdout_rowmajor_N16", lambda: renumbered_secondary(api, torch, dev, stream, "holdout", 16, 40, "rm")),
                        ("fem_4M_rcm_node_order_N16",
                         lambda: fem_secondary(api, torch, dev, stream, (110, 110, 110, 3), 16, 40, numbering="rcm")),
                        ("holdout_kron_nasa4704_4M_N16", lambda: holdout_secondary(api, torch, dev, stream, 16, 40)),
                        ("rowmajor_holdout_kron_nasa4704_4M_N16", lambda: holdout_secondary(api, torch, dev, stream, 16, 40, layout="rm")),
                        ("rowmajor_holdout_kron_nasa4704_4M_random_order_N16",
                         lambda: holdout_secondary(api, torch, dev, stream, 16, 40, numbering="random", layout="rm")),
                        ("holdout_kron_nasa4704_4M_random_order_N16",
                         lambda: holdout_secondary(api, torch, dev, stream, 16, 40, numbering="random")),
                        ("holdout_kron_nasa4704_4M_rect_N16", lambda: holdout_secondary(api, torch, dev, stream, 16, 40, variant="rect")),
                        ("holdout_kron_nasa4704_4M_unsym_N16", lambda: holdout_secondary(api, torch, dev, stream, 16, 40, variant="unsym")),
                        ("fem27pt_1dof_4M_N16", lambda: fem_secondary(api, torch, dev, stream, (160, 160, 160, 1), 16, 50)),
                        ("stencil2d_5pt_4M_N16", lambda: stencil_secondary(api, torch, dev, stream, 2000, 2000, 5, 16, 50)),
                        ("rowmajor_stencil2d_5pt_4M_N16", lambda: stencil_secondary(api, torch, dev, stream, 2000, 2000, 5, 16, 50, layout="rm")),
                        ("rowmajor_fem27pt_1dof_4M_N16", lambda: fem_secondary(api, torch, dev, stream, (160, 160, 160, 1), 16, 50, layout="rm")),
                        ("stencil2d_9pt_4M_N16", lambda: stencil_secondary(api, torch, dev, stream, 2000, 2000, 9, 16, 50)),
                        ("config5_blocked_ell_bf16_N256", lambda: bell_secondary(api, torch, dev, stream)),
                        ("blockbanded_ell_bf16_N256", lambda: bell_secondary(api, torch, dev, stream, banded_half_width=127)),
                        ("powerlaw_1M_rows_N16", lambda: powerlaw_secondary(api, torch, dev, stream)),
                        ("kkt_3M_rows_N16", lambda: kkt_secondary(api, torch, dev, stream, 16)),
                        ("rowmajor_kkt_3M_rows_N16", lambda: kkt_secondary(api, torch, dev, stream, 16, layout="rm")),
                        ("rowmajor_kkt_3M_rows_N128", lambda: kkt_secondary(api, torch, dev, stream, 128, layout="rm")),
                        ("modes_strict_vs_fast_powerlaw_N16", lambda: modes_secondary(api, torch, dev, stream, "powerlaw", 16)),
                        ("modes_strict_vs_fast_kkt_N16", lambda: modes_secondary(api, torch, dev, stream, "kkt", 16)),
                        ("modes_strict_vs_fast_fem_4M_N128", lambda: modes_secondary(api, torch, dev, stream, "fem", 128)),
                        ("dense_blocks_fp32_mfma_vs_valu_N128", lambda: dense_blocks_secondary(api, torch, dev, stream, 128)),
                        ("config4_matrix_N32", lambda: uniform_secondary(api, torch, dev, stream, args, 32)),
                        ("rowmajor_config4_matrix_N16", lambda: uniform_secondary(api, torch, dev, stream, args, 16, layout="rm")),
                        ("rowmajor_config4_matrix_N32", lambda: uniform_secondary(api, torch, dev, stream, args, 32, layout="rm")))


def cpu_baseline(api, M, K, N, args, Cout, flops_per_row, all_cores=True, rowmajor=False):
    """Single-thread CPU baseline on the first R rows of the same matrix (same B, same C_in):
    the reference's own cpu_spmm_CSR when oracle/_ref is present, else our C restatement.
    Also cross-checks the GPU result on those rows bit-for-bit."""
    from oracle.bindings import Oracle, Ref
    o = Oracle()
    kind, ref = "port", None
    try:
        ref = Ref()
        kind = "reference"
    except Exception:
        pass
    Bh = api.gen_uniform_host(K * N, 41)
    Cin_h = api.gen_uniform_host(M * N, 42)      # element i of stream 42 is C_in[i] (column-major M x N)
    if rowmajor:   # (--rowmajor: element i is B[k][n] / C_in[r][n] with i = k N + n -- the CPU loops want column-major copies)
        Bh = np.ascontiguousarray(Bh.reshape(K, N).T).reshape(-1)
        Cin_h = np.ascontiguousarray(Cin_h.reshape(M, N).T).reshape(-1)
    cores = 1

    def run(R):
        rp, ci, v = api.gen_csr_host(M, K, args.mean_nnz, 4, 0, R)
        # rows [0,R) as an R x K problem: C sample = first R rows of every column of C_in
        Cs = np.ascontiguousarray(Cin_h.reshape(N, M)[:, :R]).reshape(-1)
        if ref is not None:
            sec = ref.spmm(R, N, K, np.float32(ALPHA), rp, ci, v, Bh, np.float32(BETA), Cs)
        else:
            sec = o.time_spmm_rows(0, R, R, N, K, np.float32(ALPHA), rp, ci, v, Bh, np.float32(BETA), Cs)
        return sec, int(rp[-1]), Cs, (rp, ci, v)

    R = min(M, 50_000)
    sec, nnz_s, Cs, csr = run(R)
    if sec < args.cpu_seconds / 4 and R < M:
        R = int(min(M, max(R, R * args.cpu_seconds / max(sec, 1e-3))))
        sec, nnz_s, Cs, csr = run(R)
    gf = 2.0 * N * (nnz_s + R) / sec / 1e9
    got = (Cout.view(M, N)[:R].t().contiguous() if rowmajor else Cout.view(N, M)[:, :R]).cpu().numpy().reshape(-1)
    match = bool(np.array_equal(got.view(np.uint32), Cs.view(np.uint32)))
    out = {"value": round(gf, 3), "unit": "GFLOP/s", "cores": cores, "kind": kind,
           "sample": f"rows [0,{R}) of the same matrix ({nnz_s} nnz), same B and C_in, "
                     f"{sec:.2f} s single thread; host has {os.cpu_count()} logical cores",
           "gpu_matches_cpu_bitwise_on_sample": match}
    if not all_cores:
        return out
    # Same loop nest, row-parallel on ALL host cores (SURVEY.md 8d baseline 2): OpenMP loop in oracle/, same sample.
    try:
        rp, ci, v = csr
        Cs2 = np.ascontiguousarray(Cin_h.reshape(N, M)[:, :R]).reshape(-1)
        o.time_spmm_omp(R, N, K, np.float32(ALPHA), rp, ci, v, Bh, np.float32(BETA), Cs2.copy())   # thread start-up
        sec2, threads = o.time_spmm_omp(R, N, K, np.float32(ALPHA), rp, ci, v, Bh, np.float32(BETA), Cs2)
        out["all_cores"] = {"value": round(2.0 * N * (nnz_s + R) / sec2 / 1e9, 3), "unit": "GFLOP/s",
                            "cores": threads, "kind": "port", "seconds": round(sec2, 3),
                            "host_logical_cores": os.cpu_count(),
                            "matches_single_thread_bitwise": bool(np.array_equal(Cs2.view(np.uint32),
                                                                                  Cs.view(np.uint32)))}
    except Exception as e:   # extra information only
        out["all_cores"] = {"error": str(e)}
    return out


def _measure(api, torch, e, M, K, N, nnz, dev, stream, iters, layout="cm"):
    """Steady-state step time (wall, synchronised) and HIP-event time of the dominant kernel.
    layout "rm": the same operands read as ROW-major K x N / M x N through sextans_spmm_device_rm (no layout passes)."""
    B = torch.empty(K * N, dtype=torch.float32, device=dev)
    Cin = torch.empty(M * N, dtype=torch.float32, device=dev)
    Cout = torch.empty(M * N, dtype=torch.float32, device=dev)
    api.gen_uniform_device(dev.index, B.data_ptr(), K * N, 41, stream)
    api.gen_uniform_device(dev.index, Cin.data_ptr(), M * N, 42, stream)
    f = lambda: e.spmm_device(N, ALPHA, B.data_ptr(), K, BETA, Cin.data_ptr(), Cout.data_ptr(), M, stream)
    if layout == "rm":
        f = lambda: e.spmm_device_rm(N, ALPHA, B.data_ptr(), N, BETA, Cin.data_ptr(), N, Cout.data_ptr(), N, stream)
    # warm-up until the device has run this workload for ~60 ms: the first ~30 launches after an idle phase run 5-8 % slower (clock
    # ramp; tools/placement.py, tools/thermal.py: 593 us for the first 30 kernels of the FEM matrix, 549 afterwards, flat for 8 s)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f()
    torch.cuda.synchronize()
    one = max(time.perf_counter() - t0, 1e-6)
    for _ in range(min(300, int(0.06 / one))):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        f()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / iters
    e.set_option("profile", 1); e.profile_reset()
    for _ in range(min(iters, 20)):
        f()
    torch.cuda.synchronize()
    k_ns, _, rp_ns = e.profile_read()
    post_ns, _ = e.profile_read_post()
    e.set_option("profile", 0); e.profile_reset()
    by = alg_bytes(M, K, N, nnz)
    out = {"M": M, "K": K, "N": N, "nnz": nnz, "kernel": e.last_kernel(),
           "us_per_step": round(per * 1e6, 2), "gflops": round(2.0 * N * (nnz + M) / per / 1e9, 1),
           "kernel_us": round(k_ns / 1e3, 2), "repack_us": round(rp_ns / 1e3, 2),
           "plan_build_s": round(e.get_stat("plan_build_s"), 3),
           "alg_gbs_kernel": round(by / (k_ns * 1e-9) / 1e9, 1),
           "roofline_frac_kernel": round(by / (k_ns * 1e-9) / 1e9 / HBM_PEAK_GBS, 4),
           "roofline_frac_step": round(by / per / 1e9 / HBM_PEAK_GBS, 4)}
    if layout == "rm":
        out["layout"] = "row-major B and C (sextans_spmm_device_rm)"
    if post_ns:      # the reordered form: "repack_us" = B into permuted panels + C_in into the staging buffer, this = staging -> C_out
        out["post_us"] = round(post_ns / 1e3, 2)
    state = int(e.get_stat("row_cluster"))
    if state > 0:
        out["row_order"] = {1: "runs of 16 rows clustered over the graph of runs" if e.get_stat("cluster_runs") else "grid bricks",
                            2: "graph clustering"}[state]
        out["panel_rows_natural"] = int(e.get_stat("panel_rows_natural"))
        out["panel_rows_clustered"] = int(e.get_stat("panel_rows_clustered"))
    return out


def nasa_secondary(api, torch, dev, stream, layout="cm"):
    """BASELINE config 2: nasa4704.mtx, N=16 -- latency-bound and cache-resident; per-launch mean
    over 1000 back-to-back steps."""
    path = os.path.join(ROOT, "matrices", "nasa4704", "nasa4704.mtx")
    rp, ci, v, M, K, nnz = api.read_suitsparse_matrix(path)
    e = api.Engine(dev.index)
    e.set_matrix_csr(M, K, rp, ci, v)
    out = _measure(api, torch, e, M, K, 16, nnz, dev, stream, 1000, layout)
    if layout == "rm":   # (eager steps on the row-major entry point: the LDS-panel kernel reads the caller's B by LDS-DMA, no 4-byte staging loads)
        e.close()
        return out
    # the reference's own protocol: `sextans nasa4704.mtx 16 <rp_time>` = rp_time repeats of the kernel on
    # resident inputs (sextans-host.cpp:237-260); the engine replays them as one hipGraph
    Bh, Ch = api.init_dense_B(K, 16), api.init_dense_C(M, 16)
    e.spmm(16, ALPHA, Bh, BETA, Ch.copy(), rp_time=10)
    ns = e.spmm(16, ALPHA, Bh, BETA, Ch, rp_time=1000)
    out["rp_time_1000_us_per_repeat"] = round(ns / 1000 / 1e3, 3)
    out["rp_time_1000_gflops"] = round(api.gflops(M, 16, nnz, ns * 1e-9 / 1000), 1)
    # A pair of HIP events around ONE launch costs more than this kernel runs (the event figure is ~2.4x the whole step): the
    # kernel time of this workload is the per-repeat time of the hipGraph replay, back-to-back launches without host gaps.
    out["kernel_us_one_launch_between_hip_events"] = out["kernel_us"]
    k_us = ns / 1000 / 1e3
    by = alg_bytes(M, K, 16, nnz)
    out["kernel_us"] = round(k_us, 3)
    out["kernel_us_source"] = "hipGraph replay of 1000 repeats (sextans_spmm_host rp_time = 1000), per repeat"
    out["alg_gbs_kernel"] = round(by / (k_us * 1e-6) / 1e9, 1)
    out["roofline_frac_kernel"] = round(by / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
    e.close()
    return out


def bell_secondary(api, torch, dev, stream, M=1_048_576, W=328, N=256, iters=5, banded_half_width=None):
    """BASELINE config 5: blocked-ELL 32x32 bf16 blocks, 1% block fill, N=256, bf16 MFMA path.
    F = 2*N*(1024*nblocks + M); bytes = 2048*nb + 4*nb + 2*K*N + 8*M*N.
    banded_half_width: the block-banded variant instead (2*hw + 1 consecutive block columns per block row: block rows
    share block columns, the input class where a B tile staged once per workgroup is reused -- spmm_bell_mfma_shared)."""
    K = M
    if banded_half_width is not None:
        W = 2 * banded_half_width + 1
        dc, dv = api.gen_bell_banded_device(dev.index, M, K, banded_half_width, 5)
    else:
        dc, dv = api.gen_bell_device(dev.index, M, K, W, 5)
    e = api.Engine(dev.index)
    e.set_matrix_bell_device(M, K, W, dc, dv)
    api.device_free(dev.index, dv)                      # the engine keeps its own fragment-order copy
    B = torch.empty(K * N, dtype=torch.int16, device=dev)
    Cin = torch.empty(M * N, dtype=torch.float32, device=dev)
    Cout = torch.empty(M * N, dtype=torch.float32, device=dev)
    api.gen_uniform_bf16_device(dev.index, B.data_ptr(), K * N, 51, stream)
    api.gen_uniform_device(dev.index, Cin.data_ptr(), M * N, 52, stream)
    f = lambda: e.spmm_bell_device(N, ALPHA, B.data_ptr(), K, BETA, Cin.data_ptr(), Cout.data_ptr(), M, stream)
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        f()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / iters
    e.set_option("profile", 1); e.profile_reset()
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    k_ns, _, rp_ns = e.profile_read()
    nb = (M // 32) * W
    flops = 2.0 * N * (1024.0 * nb + M)
    by = 2048 * nb + 4 * nb + 2 * K * N + 8 * M * N
    out = {"M": M, "K": K, "N": N, "blocks": nb, "kernel": e.last_kernel(), "ms_per_step": round(per * 1e3, 3),
           "tflops": round(flops / per / 1e12, 1), "kernel_ms": round(k_ns / 1e6, 3),
           "repack_b_us": round(rp_ns / 1e3, 1), "alg_gbs_kernel": round(by / (k_ns * 1e-9) / 1e9, 1),
           "roofline_frac_kernel": round(by / (k_ns * 1e-9) / 1e9 / HBM_PEAK_GBS, 4),
           "mfma_util_vs_2.5PF": round(flops / (k_ns * 1e-9) / 2.5e15, 4), "dtype": "bf16 in, f32 accumulate"}
    if banded_half_width is not None:
        out["matrix"] = f"block-banded, {W} consecutive block columns per block row"
        out["blocks_per_distinct_column_in_a_workgroup"] = round(e.get_stat("bell_share"), 2)
    # rocprof MFMA utilisation (north_star): a counter figure of the same workload, stored with its provenance -- never measured in this run
    try:
        stored = json.load(open(os.path.join(ROOT, "profiles", "r06_mfma_busy.json")))
        rec = stored.get("blockbanded_ell_bf16_N256" if banded_half_width is not None else "config5_blocked_ell_bf16_N256")
        if rec and (M, W, N) in ((1_048_576, 328, 256), (1_048_576, 255, 256)):
            if rec.get("kernel") == e.last_kernel():
                out["mfma_busy_frac"] = rec["mfma_busy_frac"]
                out["mfma_busy_frac_source"] = rec["source"]
            else:
                out["mfma_busy_frac_source"] = f"NOT APPLICABLE: stored counters are for kernel {rec.get('kernel')!r}, this run used {e.last_kernel()!r}"
    except (OSError, ValueError):
        pass
    e.close()
    api.device_free(dev.index, dc)
    return out


def powerlaw_secondary(api, torch, dev, stream):
    """Skewed input of the sweep harness: 1M x 1M, P(len >= x) = (6/x)^1.2, longest row ~400 000,
    long rows bucketed (default) and hub rows split and folded in order (split_rows = -1, opt-in), next to a uniform
    matrix with the same number of non-zeros."""
    M = K = 1_000_000
    p, i, v, nnz = api.gen_powerlaw_device(dev.index, M, K, 6, 120, 400_000, 7)
    e = api.Engine(dev.index)
    e.set_option("split_rows", -1)   # opt in: hub rows are cut at T = max(1024, nnz/16384) and re-associated (engine default
                                     # since round 3 is strict order for every row)
    e.set_matrix_csr_device(M, K, nnz, p, i, v)
    out = _measure(api, torch, e, M, K, 16, nnz, dev, stream, 50)
    out.update(options="split_rows=-1 (opt-in re-association of hub rows)", piece_path_rows=int(e.get_stat("piece_path_rows")), reassociated_rows=int(e.get_stat("reassociated_rows")),
               matrix="powerlaw xmin 6, tail 1.2, max 400000, seed 7")
    e.close()
    # the same matrix with NO option set: strict CSR order for every row, the 399 302-entry row summed by an exact chain
    # (bit-identical to cpu_spmm_CSR)
    e = api.Engine(dev.index)
    e.set_matrix_csr_device(M, K, nnz, p, i, v)
    d = _measure(api, torch, e, M, K, 16, nnz, dev, stream, 30)
    out["default_strict_order"] = {"us_per_step": d["us_per_step"], "gflops": d["gflops"], "kernel": d["kernel"],
                                   "exact_chain_rows": int(e.get_stat("exact_chain_rows")),
                                   "reassociated_rows": int(e.get_stat("reassociated_rows"))}
    e.close()
    for q in (p, i, v):
        api.device_free(dev.index, q)
    p, i, v, nnz_u = api.gen_csr_device(dev.index, M, K, nnz / M, 7)
    e = api.Engine(dev.index)
    e.set_matrix_csr_device(M, K, nnz_u, p, i, v)
    u = _measure(api, torch, e, M, K, 16, nnz_u, dev, stream, 50)
    out["uniform_same_nnz_us_per_step"] = u["us_per_step"]
    out["ratio_to_uniform"] = round(out["us_per_step"] / u["us_per_step"], 3)
    e.close()
    for q in (p, i, v):
        api.device_free(dev.index, q)
    return out


def kkt_secondary(api, torch, dev, stream, N, layout="cm"):
    """KKT / arrow system (2M variables with a pentadiagonal Hessian, 1M constraints, 4 dense borders): short rows on the LDS-panel
    kernel + border rows on the long-row paths (bucketed pieces, exact chains: strict order, bit-identical).  layout "rm": the same
    through the row-major entry point, long-row kernels included (no copies)."""
    M = K = api.kkt_rows(2_000_000, 4)
    p, i, v, nnz = api.gen_kkt_device(dev.index, 2_000_000, 4, 3)
    e = api.Engine(dev.index)
    e.set_matrix_csr_device(M, K, nnz, p, i, v)
    out = _measure(api, torch, e, M, K, N, nnz, dev, stream, 30, layout)
    out.update(piece_path_rows=int(e.get_stat("piece_path_rows")), exact_chain_rows=int(e.get_stat("exact_chain_rows")))
    e.close()
    for q in (p, i, v):
        api.device_free(dev.index, q)
    return out


def modes_secondary(api, torch, dev, stream, which, N):
    """SEXTANS_MODE_STRICT (default: bit-identical to cpu_spmm_CSR) and SEXTANS_MODE_FAST ("exact" 0 + "split_rows" -1: inside
    1e-4 * (|alpha| sum|a b| + |beta c|), include/sextans_amd.h) side by side on one matrix, one engine per mode.
    which: "powerlaw" | "kkt" | "fem"."""
    if which == "powerlaw":
        M = K = 1_000_000
        p, i, v, nnz = api.gen_powerlaw_device(dev.index, M, K, 6, 120, 400_000, 7); name = "powerlaw xmin 6, tail 1.2, max 400000, seed 7"; iters = 30
    elif which == "kkt":
        M = K = api.kkt_rows(2_000_000, 4)
        p, i, v, nnz = api.gen_kkt_device(dev.index, 2_000_000, 4, 3); name = "kkt 2M variables, 1M constraints, 4 borders"; iters = 30
    else:
        M = K = 110 ** 3 * 3
        p, i, v, nnz = api.gen_fem3d_device(dev.index, 110, 110, 110, 3, 3); name = "fem3d 110x110x110, 3 dof/node"; iters = 10
    out = {"matrix": name, "N": N}
    for label, mode in (("strict", 0), ("fast", 1)):
        e = api.Engine(dev.index)
        e.set_option("mode", mode)
        e.set_matrix_csr_device(M, K, nnz, p, i, v)
        d = _measure(api, torch, e, M, K, N, nnz, dev, stream, iters)
        out[label] = {k: d[k] for k in ("kernel", "us_per_step", "kernel_us", "gflops", "roofline_frac_kernel", "roofline_frac_step")}
        out[label]["reassociated_rows"] = int(e.get_stat("reassociated_rows"))
        e.close()
    out["fast_over_strict"] = round(out["strict"]["us_per_step"] / out["fast"]["us_per_step"], 3)
    out["roofline_frac_step"] = out["strict"]["roofline_frac_step"]          # (the headline figure of the entry is the default mode's)
    out["roofline_frac_kernel"] = out["strict"]["roofline_frac_kernel"]
    out["guarantee_fast"] = "|C_fast - C_ref| <= 1e-4 * (|alpha| sum|a b| + |beta c|) per element; tests/test_rowblock_mfma_gpu.py::test_mode_switch_and_its_guarantee"
    for q in (p, i, v):
        api.device_free(dev.index, q)
    return out


def dense_blocks_secondary(api, torch, dev, stream, N):
    """"MFMA only where a tile is actually dense", fp32: kron(T_32768, dense 32 x 32) -- block tridiagonal, fully dense blocks, 1 M rows,
    100.6 M non-zeros -- on the VALU kernels ("exact" = 0) and with its row blocks routed to v_mfma_f32_16x16x4_f32 ("mfma_dense_tiles" = 2):
    the SAME bits (tests/test_rowblock_mfma_gpu.py), side by side."""
    import numpy as np
    bs = 32
    prp = (np.arange(bs + 1) * bs).astype(np.int32); pci = np.tile(np.arange(bs, dtype=np.int32), bs)
    n = 32768
    p, i, v, nnz, K = api.gen_kron_device(dev.index, n, prp, pci, bs, 0, 7)
    M = n * bs
    out = {"matrix": "kron(T_32768, dense 32x32): block tridiagonal, fully dense fp32 blocks", "N": N}
    for label, opts in (("valu_exact0", {"exact": 0}), ("mfma_f32_row_blocks", {"exact": 0, "mfma_dense_tiles": 2})):
        e = api.Engine(dev.index)
        for k, val in opts.items():
            e.set_option(k, val)
        e.set_matrix_csr_device(M, K, nnz, p, i, v)
        d = _measure(api, torch, e, M, K, N, nnz, dev, stream, 20)
        out[label] = {k: d[k] for k in ("kernel", "us_per_step", "kernel_us", "repack_us", "gflops", "roofline_frac_kernel", "roofline_frac_step", "plan_build_s")}
        out[label]["tflops"] = round(2.0 * N * nnz / (d["us_per_step"] * 1e-6) / 1e12, 2)
        if "mfma_dense_tiles" in opts:
            out[label]["routed_fraction"] = round(e.get_stat("dense_tile_fraction"), 4)
        e.close()
    out["roofline_frac_step"] = out["valu_exact0"]["roofline_frac_step"]
    out["roofline_frac_kernel"] = out["valu_exact0"]["roofline_frac_kernel"]
    for q in (p, i, v):
        api.device_free(dev.index, q)
    return out


def uniform_secondary(api, torch, dev, stream, args, N, layout="cm"):
    """The config-4 matrix at N = 32: a B row is a whole 128-byte line, one fabric request per non-zero carries twice
    the payload of N = 16 (8 lanes per row, chosen automatically).  layout "rm": the headline matrix through the row-major entry
    point -- the gather kernel reads the caller's B rows (no repack) and writes 16 bytes of a C row per lane."""
    M = K = args.rows
    p, i, v, nnz = api.gen_csr_device(dev.index, M, K, args.mean_nnz, 4)
    e = api.Engine(dev.index)
    e.set_matrix_csr_device(M, K, nnz, p, i, v)
    out = _measure(api, torch, e, M, K, N, nnz, dev, stream, 10, layout)
    e.close()
    for q in (p, i, v):
        api.device_free(dev.index, q)
    return out


def fem_secondary(api, torch, dev, stream, dims, N, iters, numbering="grid", options=None, layout="cm", torch_op=False, rp_protocol=False):
    """SuiteSparse-like FEM input (27-point node stencil, `dof` unknowns per node): the class of
    matrices with B-row reuse, where the LDS-panel kernel applies.  dims = (nx, ny, nz, dof).
    numbering: "grid" (natural order), "random" (a seeded random renumbering of the NODES, applied in HBM: what an arbitrary
    mesh numbering looks like to the kernels) or "rcm" (reverse Cuthill-McKee of the node graph, scipy on the host)."""
    nx, ny, nz, dof = dims
    M = K = nx * ny * nz * dof
    p, i, v, nnz = api.gen_fem3d_device(dev.index, nx, ny, nz, dof, 3)
    if numbering != "grid":
        from sextans_amd import meshgen
        if numbering == "random":
            perm = meshgen.node_permutation(M // dof, dof, 1)
        else:
            rp1, ci1, _ = api.gen_fem3d_host(nx, ny, nz, 1, 3)
            perm = meshgen.expand_dof(meshgen.rcm_node_permutation(rp1, ci1, nx * ny * nz, 1), dof)
        q = api.permute_symmetric_device(dev.index, M, nnz, p, i, v, perm)
        for old in (p, i, v):
            api.device_free(dev.index, old)
        p, i, v = q
    e = api.Engine(dev.index)
    for k, val in (options or {}).items():
        e.set_option(k, val)
    e.set_matrix_csr_device(M, K, nnz, p, i, v)
    out = _measure(api, torch, e, M, K, N, nnz, dev, stream, iters, layout)
    out["matrix"] = f"fem3d {nx}x{ny}x{nz}, {dof} dof/node" + ("" if numbering == "grid" else f", {numbering} node order")
    if options:
        out["options"] = options
    if rp_protocol:   # the reference's own protocol: `sextans <A.mtx> <N> <rp_time>` = rp_time repeats on resident inputs (sextans-host.cpp:237-260)
        Bh, Ch = api.init_dense_B(K, N), api.init_dense_C(M, N)
        e.spmm(N, ALPHA, Bh, BETA, Ch.copy(), rp_time=10)
        ns = e.spmm(N, ALPHA, Bh, BETA, Ch, rp_time=1000)
        out["rp_time_1000_us_per_repeat"] = round(ns / 1000 / 1e3, 3)
        out["rp_time_1000_gflops"] = round(api.gflops(M, N, nnz, ns * 1e-9 / 1000), 1)
        out["rp_time_1000_kernel"] = e.last_kernel()
        out["rp_time_1000_roofline_frac"] = round(alg_bytes(M, K, N, nnz) / (ns / 1000 * 1e-9) / 1e9 / HBM_PEAK_GBS, 4)
    e.close()
    if torch_op:     # the operator front end on the same matrix: wall time per call of torch_op.spmm on row-major tensors, in place
        from sextans_amd import torch_op as top
        rp_t = torch.empty(M + 1, dtype=torch.int32, device=dev); ci_t = torch.empty(nnz, dtype=torch.int32, device=dev); v_t = torch.empty(nnz, device=dev)
        for dst, src, n in ((rp_t, p, M + 1), (ci_t, i, nnz), (v_t, v, nnz)):
            api.device_copy(dev.index, dst.data_ptr(), src, 4 * n)
        A = torch.sparse_csr_tensor(rp_t, ci_t, v_t, size=(M, K))
        Bt = torch.empty((K, N), device=dev); Ct = torch.empty((M, N), device=dev); Ot = torch.empty((M, N), device=dev)
        api.gen_uniform_device(dev.index, Bt.data_ptr(), K * N, 41, stream); api.gen_uniform_device(dev.index, Ct.data_ptr(), M * N, 42, stream)
        g = lambda: top.spmm(A, Bt, ALPHA, BETA, Ct, out=Ot)
        for _ in range(100):
            g()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            g()
        torch.cuda.synchronize()
        out["torch_op_us_per_call"] = round((time.perf_counter() - t0) / iters * 1e6, 2)
        top.clear_cache()
        del A, rp_t, ci_t, v_t, Bt, Ct, Ot
    if numbering == "random":        # the same matrix on the natural-order forms (what round 3 would have run)
        e = api.Engine(dev.index)
        e.set_option("row_cluster", 0)
        e.set_matrix_csr_device(M, K, nnz, p, i, v)
        d = _measure(api, torch, e, M, K, N, nnz, dev, stream, max(3, iters // 4))
        out["without_row_clustering"] = {k: d[k] for k in ("kernel", "us_per_step", "kernel_us", "repack_us", "roofline_frac_kernel", "roofline_frac_step")}
        e.close()
    for q in (p, i, v):
        api.device_free(dev.index, q)
    return out


def holdout_secondary(api, torch, dev, stream, N, iters, n=850, variant="", numbering="natural", layout="cm"):
    """HOLDOUT class (round 5): kron(T_n, nasa4704) -- the local structure of the one real SuiteSparse matrix in this mount carried
    to 4 M rows (sextans_amd/holdout.py); the generators of the other `also` entries were written next to the dispatcher's
    heuristics, this one was not.  variant: "" | "rect" (every third column dropped) | "unsym" (30 % of the lower entries dropped)."""
    from sextans_amd import holdout
    M, K, p, i, v, nnz = holdout.kron_device(dev.index, n, variant, numbering)
    e = api.Engine(dev.index)
    e.set_matrix_csr_device(M, K, nnz, p, i, v)
    out = _measure(api, torch, e, M, K, N, nnz, dev, stream, iters, layout)
    out["matrix"] = f"kron(T_{n}, nasa4704)" + (f", {variant}" if variant else "") + f", {numbering} numbering"
    out["cluster_decline"] = int(e.get_stat("cluster_decline"))
    e.close()
    for q in (p, i, v):
        api.device_free(dev.index, q)
    return out


def renumbered_secondary(api, torch, dev, stream, which, N, iters, layout="cm"):
    """A matrix RENUMBERED ONCE by the engine's own clustered row order (sextans_export_row_order; P A P^T in HBM, what FEM packages do
    with RCM -- the caller permutes the rows of B and C the same way): its natural-order forms then run without any layout pass and
    its contiguous row ranges are compact pieces of the graph.  which: "fem_random" | "holdout"."""
    if which == "fem_random":
        from sextans_amd import meshgen
        M = K = 110 * 110 * 110 * 3
        p, i, v, nnz = api.gen_fem3d_device(dev.index, 110, 110, 110, 3, 3)
        q = api.permute_symmetric_device(dev.index, M, nnz, p, i, v, meshgen.node_permutation(M // 3, 3, 1))
        for old in (p, i, v):
            api.device_free(dev.index, old)
        p, i, v = q
        name = "fem3d 110x110x110, 3 dof/node, random node order"
    else:
        from sextans_amd import holdout
        M, K, p, i, v, nnz = holdout.kron_device(dev.index, 850)
        name = "kron(T_850, nasa4704), as generated"
    e = api.Engine(dev.index)
    e.set_option("row_cluster", 2)
    e.set_matrix_csr_device(M, K, nnz, p, i, v)
    order, kind = e.export_row_order()
    e.close()
    new_of_old = np.empty(M, np.int64)
    new_of_old[order] = np.arange(M)
    q = api.permute_symmetric_device(dev.index, M, nnz, p, i, v, new_of_old)
    for old in (p, i, v):
        api.device_free(dev.index, old)
    p, i, v = q
    e = api.Engine(dev.index)
    e.set_matrix_csr_device(M, K, nnz, p, i, v)
    out = _measure(api, torch, e, M, K, N, nnz, dev, stream, iters, layout)
    out["matrix"] = name + f", renumbered once by sextans_export_row_order (kind {kind}): P A P^T"
    out["cluster_decline"] = int(e.get_stat("cluster_decline"))
    e.close()
    for x in (p, i, v):
        api.device_free(dev.index, x)
    return out


def stencil_secondary(api, torch, dev, stream, nx, ny, points, N, iters, layout="cm"):
    """2-D stencil on an nx x ny grid (short rows: 5 or 9 entries): 5-point runs on spmm_csr_colwise (no B repack)."""
    M = K = nx * ny
    p, i, v, nnz = api.gen_stencil2d_device(dev.index, nx, ny, points, 1, 3)
    e = api.Engine(dev.index)
    e.set_matrix_csr_device(M, K, nnz, p, i, v)
    out = _measure(api, torch, e, M, K, N, nnz, dev, stream, iters, layout)
    out["matrix"] = f"2-D {points}-point stencil {nx}x{ny}"
    e.close()
    for q in (p, i, v):
        api.device_free(dev.index, q)
    return out


if __name__ == "__main__":
    main()
