#!/usr/bin/env python3
"""tools/collect_r06.py -- turn what tools/evidence_r06.sh / tools/sweep_r06.sh / bench.py left under gpurun_out/ into the tracked
round-6 evidence files under profiles/ (PMC files get a derived header: HBM traffic of the SpMM kernel per launch against the
algorithmic bytes of the workload)."""
import glob, json, os, re, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out") + "/", os.path.join(ROOT, "profiles") + "/"
head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()


def cp(src, dst):
    if os.path.exists(G + src):
        shutil.copy(G + src, P + dst)
        print("profiles/" + dst)


cp("r06_bench_kernel_stats_per_workload.csv", "r06_bench_kernel_stats_per_workload.csv")
cp("prof_r06/kernel_stats.csv", "r06_bench_config4_kernel_stats.csv")
cp("r06_sweep.jsonl", "r06_sweep.jsonl")
cp("r06_bench_final.json", "r06_bench_line.json")

# config 4: PMC rows of the headline kernel + traffic JSON (what bench.py reports as roofline.traffic)
if os.path.isdir(G + "prof_r06"):
    txt, vals = "", {}
    for f in sorted(glob.glob(G + "prof_r06/pmc_*.txt")):
        for l in open(f):
            if "rowgroup" in l or l.startswith("kernel"):
                txt += l
            m = re.search(r"rowgroup.*\s(\S+)\s+(\d+)\s+([\d.]+)\s*$", l)
            if m:
                vals[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    open(P + "r06_bench_config4_pmc.txt", "w").write(
        "# tools/prof.sh r06 at commit %s: separate rocprofv3 --pmc passes of\n# `python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also` (headline kernel rows only)\n" % head + txt)
    print("profiles/r06_bench_config4_pmc.txt")
    if "TCC_EA0_RDREQ_128B_sum" in vals and "WRITE_SIZE" in vals:
        rd = vals["TCC_EA0_RDREQ_128B_sum"][1] * 128 + vals["TCC_EA0_RDREQ_64B_sum"][1] * 64 + vals["TCC_EA0_RDREQ_32B_sum"][1] * 32
        wr = vals["WRITE_SIZE"][1] * 1024
        json.dump({"kernel": "spmm_csr_rowgroup", "kernel_instance": "spmm_csr_rowgroup<4,2048,true,true>", "workload": "config4 (bench.py default)",
                   "launches": vals["FETCH_SIZE"][0], "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
                   "tcc_ea0_rdreq_128b": vals["TCC_EA0_RDREQ_128B_sum"][1], "tcc_ea0_rdreq_64b": vals["TCC_EA0_RDREQ_64B_sum"][1],
                   "fetch_size_kb": vals["FETCH_SIZE"][1], "fetch_size_x2_bytes": vals["FETCH_SIZE"][1] * 2048, "write_size_kb": vals["WRITE_SIZE"][1],
                   "tcc_hit": vals.get("TCC_HIT_sum", (0, 0))[1], "tcc_miss": vals.get("TCC_MISS_sum", (0, 0))[1],
                   "source": "tools/prof.sh r06 at commit %s: separate rocprofv3 --pmc passes of `python bench.py --steps 10 --warmup 2 --no-cpu-baseline "
                             "--no-also` (profiles/r06_bench_config4_pmc.txt); reads = TCC_EA0_RDREQ_128B x 128 B + _64B x 64 B (= 2 x FETCH_SIZE, the "
                             "gfx950 correction of MI355X_MICROARCH.md), writes = WRITE_SIZE" % head}, open(P + "r06_config4_traffic.json", "w"), indent=1)
        print("profiles/r06_config4_traffic.json")

cp("r06_bench_forced_dist_colmajor.json", "r06_bench_forced_dist_colmajor.json")
cp("r06_bench_forced_dist_rowmajor.json", "r06_bench_forced_dist_rowmajor.json")
cp("r06_bench_stdout.txt", "r06_bench_stdout_two_lines.txt")
