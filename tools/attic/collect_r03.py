#!/usr/bin/env python3
"""tools/collect_r03.py -- turn what tools/prof_r03.sh left under gpurun_out/ into the tracked round-3 evidence files under profiles/
(kernel stats of the bench, config-4 PMC + traffic JSON, FEM N=16 / N=128 PMC + power, block-banded MFMA counters + power)."""
import glob, json, os, re, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G, P = os.path.join(ROOT, "gpurun_out") + "/", os.path.join(ROOT, "profiles") + "/"
head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()

shutil.copy(G + "prof_r03/kernel_stats.csv", P + "r03_bench_config4_kernel_stats.csv")
txt, vals = "", {}
for f in sorted(glob.glob(G + "prof_r03/pmc_*.txt")):
    for l in open(f):
        if "rowgroup" in l or l.startswith("kernel"):
            txt += l
        m = re.search(r"rowgroup.*\s(\S+)\s+(\d+)\s+([\d.]+)\s*$", l)
        if m:
            vals[m.group(1)] = (int(m.group(2)), float(m.group(3)))
open(P + "r03_bench_config4_pmc.txt", "w").write(
    "# tools/prof.sh r03 at commit %s: separate rocprofv3 --pmc passes of\n# `python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also` (headline kernel rows only)\n" % head + txt)
rd = vals["TCC_EA0_RDREQ_128B_sum"][1] * 128 + vals["TCC_EA0_RDREQ_64B_sum"][1] * 64 + vals["TCC_EA0_RDREQ_32B_sum"][1] * 32
wr = vals["WRITE_SIZE"][1] * 1024
json.dump({"kernel": "spmm_csr_rowgroup", "kernel_instance": "spmm_csr_rowgroup<4,2048,true,true>", "workload": "config4 (bench.py default)",
           "launches": vals["FETCH_SIZE"][0], "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
           "tcc_ea0_rdreq_128b": vals["TCC_EA0_RDREQ_128B_sum"][1], "tcc_ea0_rdreq_64b": vals["TCC_EA0_RDREQ_64B_sum"][1],
           "fetch_size_kb": vals["FETCH_SIZE"][1], "fetch_size_x2_bytes": vals["FETCH_SIZE"][1] * 2048, "write_size_kb": vals["WRITE_SIZE"][1],
           "tcc_hit": vals["TCC_HIT_sum"][1], "tcc_miss": vals["TCC_MISS_sum"][1],
           "source": "tools/prof.sh r03 at commit %s: separate rocprofv3 --pmc passes of `python bench.py --steps 10 --warmup 2 --no-cpu-baseline "
                     "--no-also` (profiles/r03_bench_config4_pmc.txt); reads = TCC_EA0_RDREQ_128B x 128 B + _64B x 64 B (= 2 x FETCH_SIZE, the "
                     "gfx950 correction of MI355X_MICROARCH.md), writes = WRITE_SIZE" % head}, open(P + "r03_config4_traffic.json", "w"), indent=1)

def load(d, pat):
    v = {}
    for l in open(G + d + "/summary.txt"):
        if re.search(pat, l):
            f = l.split(); v[f[-3]] = float(f[-1])
    return v

M, nnz = 3993000, 317587968
for tag, d, N in (("n16", "pmc_r03_fem_n16", 16), ("n128", "pmc_r03_fem_n128", 128)):
    v = load(d, "panel_v2")
    alg = 8 * nnz + 4 * (M + 1) + 4 * M * N + 8 * M * N
    rd, wr, cyc = v["FETCH_SIZE"] * 2048, v["WRITE_SIZE"] * 1024, v["GRBM_GUI_ACTIVE"] / 8
    hdr = f"""# FEM 4M (fem3d 110x110x110, 3 dof/node; M=K={M}, nnz={nnz}), N={N}: kernel spmm_csr_panel_v2 (+ repack_b_panels)
# tools/pmc.sh (one rocprofv3 --pmc pass per counter set, --kernel-trace only) around tools/run_one.py, commit {head}
# derived for spmm_csr_panel_v2, per launch:
#   HBM reads  = 2 x FETCH_SIZE KB (gfx950 correction) = {rd/1e9:.3f} GB   writes = WRITE_SIZE KB = {wr/1e9:.3f} GB   total {(rd+wr)/1e9:.3f} GB
#   algorithmic bytes (8 nnz + 4(M+1) + 4KN + 8MN)    = {alg/1e9:.3f} GB   traffic / algorithmic = {(rd+wr)/alg:.3f}
#   L2: hit {v['TCC_HIT_sum']/v['TCC_REQ_sum']*100:.1f} % of {v['TCC_REQ_sum']/1e6:.1f} M requests
#   shader cycles per XCD (GRBM_GUI_ACTIVE / 8) = {cyc/1e6:.3f} M  (kernel time x clock under the counters)
#   VALU instructions per SIMD per cycle = SQ_INSTS_VALU / 1024 / cycles = {v['SQ_INSTS_VALU']/1024/cyc:.3f}
#   LDS instructions per CU per cycle    = SQ_INSTS_LDS / 256 / cycles   = {v['SQ_INSTS_LDS']/256/cyc:.3f}
#   SQ_ACTIVE_INST_VALU x4 / (1024 SIMD x cycles) = {v['SQ_ACTIVE_INST_VALU']*4/1024/cyc:.3f}  (the counter charges one quad-cycle per VALU instruction; plain f32 VALU
#      measured at ~2 cycles per wave64 instruction per SIMD with 4 waves resident -- profiles/r03_valu_issue_microbench.txt)
#   SQ_ACTIVE_INST_LDS x4 / (256 CU x cycles) = {v['SQ_ACTIVE_INST_LDS']*4/256/cyc:.3f}  LDS array busy (ds_read_b128 = 4 LDS cycles per wave instruction, 256 B/clk/CU)
#   LDS bank conflict cycles / LDS active = {v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_ACTIVE_INST_LDS']*4,1):.4f}
#   occupancy: SQ_WAVE_CYCLES x4 / (1024 x cycles) = {v['SQ_WAVE_CYCLES']*4/1024/cyc:.2f} waves per SIMD
# power probe (tools/power_probe.sh, rocm-smi every 0.5 s while the kernel loops): see the tail of this file
"""
    open(P + f"r03_fem_{tag}_pmc.txt", "w").write(hdr + open(G + d + "/summary.txt").read() + "\n# ---- power probe: (sclk) package-W gpu-use% ----\n"
                                                  + open(G + f"power_r03_fem_{tag}.txt").read())
v = load("pmc_r03_bell_banded", "bell_mfma")
cyc, rd, wr = v["GRBM_GUI_ACTIVE"] / 8, v["FETCH_SIZE"] * 2048, v["WRITE_SIZE"] * 1024
run = [l for l in open(G + "power_r03_bell_banded.txt") if l.startswith("bell M=")]
hdr = f"""# block-banded blocked-ELL bf16, M=K=1048576, 32x32 blocks, band half width 127 blocks (W=255 blocks per block row), N=256
# kernel spmm_bell_mfma_shared (union-walk: 8 block rows per workgroup share one B tile ring); tools/pmc.sh around tools/run_bell.py, commit {head}
# derived per launch:
#   SQ_INSTS_MFMA = {v['SQ_INSTS_MFMA']/1e6:.2f} M v_mfma_f32_32x32x16_bf16 (32768 flop each) = {v['SQ_INSTS_MFMA']*32768/1e12:.3f} TFLOP
#   SQ_VALU_MFMA_BUSY_CYCLES = {v['SQ_VALU_MFMA_BUSY_CYCLES']/1e9:.3f} G (= 32 x SQ_INSTS_MFMA: 32 cycles per instruction per SIMD)
#   shader cycles per XCD (GRBM_GUI_ACTIVE / 8) = {cyc/1e6:.3f} M  -> MFMA pipe busy = BUSY_CYCLES / (1024 SIMD x cycles) = {v['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc*100:.1f} % of the cycles the chip ran
#   (the chip runs at ~1.83 GHz under this kernel -- power probe below: ~1390 W package -- so against the 2.5 PFLOP/s dense bf16 peak at 2.4 GHz
#    the same run is: {run[-1].strip() if run else 'see below'})
#   HBM reads = 2 x FETCH_SIZE KB = {rd/1e9:.2f} GB, writes {wr/1e9:.2f} GB; the A stream alone is 17.1 GB (bf16 32x32 blocks)
#   L2 hit {v['TCC_HIT_sum']/v['TCC_REQ_sum']*100:.1f} %
#   LDS: {v['SQ_INSTS_LDS']/1e6:.1f} M instr, bank-conflict cycles / LDS active cycles = {v['SQ_LDS_BANK_CONFLICT']/(v['SQ_ACTIVE_INST_LDS']*4):.4f}
"""
open(P + "r03_bell_banded_pmc.txt", "w").write(hdr + open(G + "pmc_r03_bell_banded/summary.txt").read() + "\n# ---- power probe: (sclk) package-W gpu-use% ----\n"
                                               + open(G + "power_r03_bell_banded.txt").read())
print("profiles/r03_* written from gpurun_out/ at", head)
