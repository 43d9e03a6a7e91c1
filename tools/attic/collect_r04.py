#!/usr/bin/env python3
"""tools/collect_r04.py -- turn what tools/evidence_r04.sh / tools/sweep_r04.sh / the experiment scripts left under gpurun_out/ into
the tracked round-4 evidence files under profiles/."""
import glob, json, os, re, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G, P = os.path.join(ROOT, "gpurun_out") + "/", os.path.join(ROOT, "profiles") + "/"
head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()


def cp(src, dst):
    if os.path.exists(G + src):
        shutil.copy(G + src, P + dst)
        print("profiles/" + dst)


cp("r04_bench_full_kernel_stats.csv", "r04_bench_full_kernel_stats.csv")
cp("prof_r04/kernel_stats.csv", "r04_bench_config4_kernel_stats.csv")
cp("r04_reordered_n16_kernel_stats.csv", "r04_reordered_fem_n16_kernel_stats.csv")
cp("r04_rank_slab_times.json", "r04_rank_slab_times.json")
cp("r04_sweep.jsonl", "r04_sweep.jsonl")
cp("r04_renumbered_classes.jsonl", "r04_renumbered_classes.jsonl")
for tag, what in (("a", "pipelining_fma_brickshapes"), ("b", "colmajor_staging_large_b"), ("c", "colwise_vs_auto")):
    if os.path.exists(G + f"r04_exp_{tag}.txt"):
        txt = "".join(l for l in open(G + f"r04_exp_{tag}.txt") if "amdgpu.ids" not in l)
        open(P + f"r04_experiment_{what}.txt", "w").write(f"# tools/exp_r04{tag}.sh (same-box A/B, tools/ab_opts.py: kernel / layout passes / wall us per step)\n" + txt)
        print("profiles/r04_experiment_%s.txt" % what)

# config 4: PMC rows of the headline kernel + traffic JSON (what bench.py reports as roofline.traffic)
if os.path.isdir(G + "prof_r04"):
    txt, vals = "", {}
    for f in sorted(glob.glob(G + "prof_r04/pmc_*.txt")):
        for l in open(f):
            if "rowgroup" in l or l.startswith("kernel"):
                txt += l
            m = re.search(r"rowgroup.*\s(\S+)\s+(\d+)\s+([\d.]+)\s*$", l)
            if m:
                vals[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    open(P + "r04_bench_config4_pmc.txt", "w").write(
        "# tools/prof.sh r04 at commit %s: separate rocprofv3 --pmc passes of\n# `python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also` (headline kernel rows only)\n" % head + txt)
    if "TCC_EA0_RDREQ_128B_sum" in vals and "WRITE_SIZE" in vals:
        rd = vals["TCC_EA0_RDREQ_128B_sum"][1] * 128 + vals["TCC_EA0_RDREQ_64B_sum"][1] * 64 + vals["TCC_EA0_RDREQ_32B_sum"][1] * 32
        wr = vals["WRITE_SIZE"][1] * 1024
        json.dump({"kernel": "spmm_csr_rowgroup", "kernel_instance": "spmm_csr_rowgroup<4,2048,true,true>", "workload": "config4 (bench.py default)",
                   "launches": vals["FETCH_SIZE"][0], "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
                   "tcc_ea0_rdreq_128b": vals["TCC_EA0_RDREQ_128B_sum"][1], "tcc_ea0_rdreq_64b": vals["TCC_EA0_RDREQ_64B_sum"][1],
                   "fetch_size_kb": vals["FETCH_SIZE"][1], "fetch_size_x2_bytes": vals["FETCH_SIZE"][1] * 2048, "write_size_kb": vals["WRITE_SIZE"][1],
                   "tcc_hit": vals["TCC_HIT_sum"][1], "tcc_miss": vals["TCC_MISS_sum"][1],
                   "source": "tools/prof.sh r04 at commit %s: separate rocprofv3 --pmc passes of `python bench.py --steps 10 --warmup 2 --no-cpu-baseline "
                             "--no-also` (profiles/r04_bench_config4_pmc.txt); reads = TCC_EA0_RDREQ_128B x 128 B + _64B x 64 B (= 2 x FETCH_SIZE, the "
                             "gfx950 correction of MI355X_MICROARCH.md), writes = WRITE_SIZE" % head}, open(P + "r04_config4_traffic.json", "w"), indent=1)
        print("profiles/r04_config4_traffic.json")


def load(d, pat):
    v = {}
    for l in open(G + d + "/summary.txt"):
        if re.search(pat, l):
            f = l.split()
            v[f[-3]] = float(f[-1])
    return v


M, nnz = 3993000, 317587968
for d, out, pat, N, title in (("pmc_r04_reordered_n16", "r04_reordered_fem_n16_pmc.txt", r"panel_v2<1, 6, true, false, false, 9, true", 16,
                               "fem3d 110^3 x 3 dof under a RANDOM NODE ORDER: spmm_csr_panel_v2 in its reordered form (graph-clustered plan)"),
                              ("pmc_r04_fem_n128", "r04_fem_n128_pmc.txt", r"panel_v2", 128, "fem3d 110^3 x 3 dof, natural order, grid-brick plan"),
                              ("pmc_r04_fem_n16", "r04_fem_n16_pmc.txt", r"panel_v2", 16, "fem3d 110^3 x 3 dof, natural order, grid-brick plan")):
    if not os.path.exists(G + d + "/summary.txt"):
        continue
    v = load(d, pat)
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    alg = 8 * nnz + 4 * (M + 1) + 4 * M * N + 8 * M * N
    rd, wr = v["FETCH_SIZE"] * 2048, v["WRITE_SIZE"] * 1024
    hdr = f"# {title}; M=K={M}, nnz={nnz}, N={N}\n# tools/pmc.sh (one rocprofv3 --pmc pass per counter set, --kernel-trace only), commit {head}\n" \
          f"# derived for the SpMM kernel, per launch:\n#   HBM reads = 2 x FETCH_SIZE KB (gfx950 correction) = {rd/1e9:.3f} GB   writes = WRITE_SIZE KB = {wr/1e9:.3f} GB   total {(rd+wr)/1e9:.3f} GB\n" \
          f"#   algorithmic bytes of the whole SpMM (8 nnz + 4(M+1) + 4KN + 8MN) = {alg/1e9:.3f} GB   kernel traffic / algorithmic = {(rd+wr)/alg:.3f}\n"
    if "TCC_REQ_sum" in v:
        hdr += f"#   L2: hit {v['TCC_HIT_sum']/max(v['TCC_REQ_sum'],1)*100:.1f} % of {v['TCC_REQ_sum']/1e6:.1f} M requests\n"
    if "GRBM_GUI_ACTIVE" in v and "SQ_INSTS_VALU" in v:
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        hdr += f"#   shader cycles per XCD = {cyc/1e6:.3f} M; VALU instructions per SIMD-cycle = {v['SQ_INSTS_VALU']/1024/cyc:.3f}; LDS array busy = {v['SQ_ACTIVE_INST_LDS']*4/256/cyc:.3f}; waves per SIMD = {v['SQ_WAVE_CYCLES']*4/1024/cyc:.2f}\n"
    open(P + out, "w").write(hdr + open(G + d + "/summary.txt").read())
    print("profiles/" + out)
print("profiles/r04_* written from gpurun_out/ at", head)
