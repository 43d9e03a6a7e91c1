#!/usr/bin/env python3
"""tools/collect_r05.py -- turn what tools/evidence_r05.sh / tools/sweep_r05.sh / bench.py left under gpurun_out/ into the tracked
round-5 evidence files under profiles/ (PMC files get a derived header: HBM traffic of the SpMM kernel per launch against the
algorithmic bytes of the workload)."""
import glob, json, os, re, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out") + "/", os.path.join(ROOT, "profiles") + "/"
head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()


def cp(src, dst):
    if os.path.exists(G + src):
        shutil.copy(G + src, P + dst)
        print("profiles/" + dst)


cp("r05_bench_kernel_stats_per_workload.csv", "r05_bench_kernel_stats_per_workload.csv")
cp("prof_r05/kernel_stats.csv", "r05_bench_config4_kernel_stats.csv")
cp("r05_sweep.jsonl", "r05_sweep.jsonl")
cp("r05_bench_final.json", "r05_bench_line.json")

# config 4: PMC rows of the headline kernel + traffic JSON (what bench.py reports as roofline.traffic)
if os.path.isdir(G + "prof_r05"):
    txt, vals = "", {}
    for f in sorted(glob.glob(G + "prof_r05/pmc_*.txt")):
        for l in open(f):
            if "rowgroup" in l or l.startswith("kernel"):
                txt += l
            m = re.search(r"rowgroup.*\s(\S+)\s+(\d+)\s+([\d.]+)\s*$", l)
            if m:
                vals[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    open(P + "r05_bench_config4_pmc.txt", "w").write(
        "# tools/prof.sh r05 at commit %s: separate rocprofv3 --pmc passes of\n# `python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also` (headline kernel rows only)\n" % head + txt)
    print("profiles/r05_bench_config4_pmc.txt")
    if "TCC_EA0_RDREQ_128B_sum" in vals and "WRITE_SIZE" in vals:
        rd = vals["TCC_EA0_RDREQ_128B_sum"][1] * 128 + vals["TCC_EA0_RDREQ_64B_sum"][1] * 64 + vals["TCC_EA0_RDREQ_32B_sum"][1] * 32
        wr = vals["WRITE_SIZE"][1] * 1024
        json.dump({"kernel": "spmm_csr_rowgroup", "kernel_instance": "spmm_csr_rowgroup<4,2048,true,true>", "workload": "config4 (bench.py default)",
                   "launches": vals["FETCH_SIZE"][0], "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
                   "tcc_ea0_rdreq_128b": vals["TCC_EA0_RDREQ_128B_sum"][1], "tcc_ea0_rdreq_64b": vals["TCC_EA0_RDREQ_64B_sum"][1],
                   "fetch_size_kb": vals["FETCH_SIZE"][1], "fetch_size_x2_bytes": vals["FETCH_SIZE"][1] * 2048, "write_size_kb": vals["WRITE_SIZE"][1],
                   "tcc_hit": vals.get("TCC_HIT_sum", (0, 0))[1], "tcc_miss": vals.get("TCC_MISS_sum", (0, 0))[1],
                   "source": "tools/prof.sh r05 at commit %s: separate rocprofv3 --pmc passes of `python bench.py --steps 10 --warmup 2 --no-cpu-baseline "
                             "--no-also` (profiles/r05_bench_config4_pmc.txt); reads = TCC_EA0_RDREQ_128B x 128 B + _64B x 64 B (= 2 x FETCH_SIZE, the "
                             "gfx950 correction of MI355X_MICROARCH.md), writes = WRITE_SIZE" % head}, open(P + "r05_config4_traffic.json", "w"), indent=1)
        print("profiles/r05_config4_traffic.json")

# per-workload PMC summaries with a derived header
ALG = lambda M, K, N, nnz: 8 * nnz + 4 * (M + 1) + 4 * K * N + 8 * M * N
FEM = (3993000, 3993000, 317587968)
HOLD = (3998400, 3998400, 266918288)
for d, what, (M, K, nnz), N in (("pmc_r05_fem_n128", "fem3d 110^3 x 3 dof, grid order, N = 128, COLUMN-major entry point", FEM, 128),
                                ("pmc_r05_fem_n128_rm", "fem3d 110^3 x 3 dof, grid order, N = 128, ROW-major entry point (sextans_spmm_device_rm)", FEM, 128),
                                ("pmc_r05_fem_n16", "fem3d 110^3 x 3 dof, grid order, N = 16, column-major entry point", FEM, 16),
                                ("pmc_r05_fem_n16_rm", "fem3d 110^3 x 3 dof, grid order, N = 16, row-major entry point", FEM, 16),
                                ("pmc_r05_reordered_n16", "fem3d 110^3 x 3 dof under a RANDOM node order, N = 16, column-major entry point (reordered form, contiguous XCD placement)", FEM, 16),
                                ("pmc_r05_reordered_n16_rm", "fem3d 110^3 x 3 dof under a random node order, N = 16, row-major entry point (clustered plan, natural B)", FEM, 16),
                                ("pmc_r05_holdout_n16", "HOLDOUT kron(T_850, nasa4704), as generated, N = 16, column-major entry point", HOLD, 16),
                                ("pmc_r05_holdout_n16_rm", "HOLDOUT kron(T_850, nasa4704), as generated, N = 16, row-major entry point (clustered plan)", HOLD, 16)):
    f = G + d + "/summary.txt"
    if not os.path.exists(f):
        continue
    v = {}
    for l in open(f):
        if re.search(r"spmm_csr_panel", l):
            c = l.split()
            v[c[-3]] = float(c[-1])
    alg = ALG(M, K, N, nnz)
    hdr = [f"# {what}; M = K = {M}, nnz = {nnz}", f"# tools/evidence_r05.sh (tools/pmc.sh: one rocprofv3 --pmc pass per counter set, --kernel-trace only), commit {head}",
           "# derived for the SpMM kernel, per launch:"]
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        rd, wr = 2 * v["FETCH_SIZE"] * 1024 / 1e9, v["WRITE_SIZE"] * 1024 / 1e9
        hdr.append(f"#   HBM reads = 2 x FETCH_SIZE KB (gfx950 correction) = {rd:.3f} GB   writes = WRITE_SIZE KB = {wr:.3f} GB   total {rd + wr:.3f} GB")
        hdr.append(f"#   algorithmic bytes of the whole SpMM (8 nnz + 4(M+1) + 4KN + 8MN) = {alg / 1e9:.3f} GB   kernel traffic / algorithmic = {(rd + wr) * 1e9 / alg:.3f}")
    if "TCC_HIT_sum" in v and "TCC_REQ_sum" in v:
        hdr.append(f"#   L2: hit {100 * v['TCC_HIT_sum'] / max(v['TCC_REQ_sum'], 1):.1f} % of {v['TCC_REQ_sum'] / 1e6:.1f} M requests")
    if "GRBM_GUI_ACTIVE" in v and "SQ_INSTS_VALU" in v and "SQ_WAVE_CYCLES" in v:
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        hdr.append(f"#   shader cycles per XCD = {cyc / 1e6:.3f} M; VALU instructions per SIMD-cycle = {v['SQ_INSTS_VALU'] / (cyc * 1024):.3f}; "
                   f"LDS array busy = {v.get('SQ_ACTIVE_INST_LDS', 0) / (cyc * 1024) / 4 * 4:.3f}; waves per SIMD = {v['SQ_WAVE_CYCLES'] / (cyc * 1024) / 4:.2f}")
    out = d.replace("pmc_r05_", "r05_") + "_pmc.txt"
    open(P + out, "w").write("\n".join(hdr) + "\n" + open(f).read())
    print("profiles/" + out)
if os.path.exists(P + "r05_sweep.jsonl"):
    t = subprocess.check_output(["python", os.path.join(ROOT, "tools", "sweep_table.py"), P + "r05_sweep.jsonl"]).decode()
    open(P + "r05_sweep_table.md", "w").write(t)
    print("profiles/r05_sweep_table.md")
