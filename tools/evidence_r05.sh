#!/bin/bash
# tools/evidence_r05.sh -- the round-5 evidence run on one MI355X box.
#  1. kernel-trace statistics PER WORKLOAD: one rocprofv3 pass for the headline and one per `also` entry (python bench.py --only <key>),
#     merged into gpurun_out/r05_bench_kernel_stats_per_workload.csv with a workload column -- no row ever averages launches of
#     different workloads (the r04 file averaged N = 16 / 32 / 128 launches of one kernel name in one row);
#  2. PMC passes (separate rocprofv3 --pmc runs, --kernel-trace only): config 4 (tools/prof.sh), FEM N = 16 / 128 through the
#     column-major and the row-major entry points, the reordered form, the holdout class.
cd "$(dirname "$0")/.."
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_kernel_stats
stats() {   # stats <key> <command...>
  local key=$1; shift
  rm -rf /tmp/rp_$key
  (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$key -o s -- "$@" > $REPO/gpurun_out/r05_kernel_stats/$key.log 2>&1)
  find /tmp/rp_$key -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05_kernel_stats/$key.csv \;
}
stats headline_config4 python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also
for KEY in $(python bench.py --only list 2>/dev/null | tail -1); do stats $KEY python $REPO/bench.py --only $KEY; done
python tools/merge_kernel_stats.py gpurun_out/r05_kernel_stats > gpurun_out/r05_bench_kernel_stats_per_workload.csv
bash tools/prof.sh r05
SETS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY;SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES;SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU;GRBM_GUI_ACTIVE"
SHORT="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum;GRBM_GUI_ACTIVE"
PMC_SETS=$SETS  bash tools/pmc.sh gpurun_out/pmc_r05_fem_n128      python $REPO/tools/run_one_opts.py 110x110x110x3 128 3
PMC_SETS=$SETS  bash tools/pmc.sh gpurun_out/pmc_r05_fem_n128_rm   python $REPO/tools/run_one_opts.py 110x110x110x3 128 3 --rm
PMC_SETS=$SHORT bash tools/pmc.sh gpurun_out/pmc_r05_fem_n16       python $REPO/tools/run_one_opts.py 110x110x110x3 16 5
PMC_SETS=$SHORT bash tools/pmc.sh gpurun_out/pmc_r05_fem_n16_rm    python $REPO/tools/run_one_opts.py 110x110x110x3 16 5 --rm
PMC_SETS=$SETS  bash tools/pmc.sh gpurun_out/pmc_r05_reordered_n16 python $REPO/tools/run_one_opts.py synth:femperm:110:110:110:3:random 16 4
PMC_SETS=$SHORT bash tools/pmc.sh gpurun_out/pmc_r05_reordered_n16_rm python $REPO/tools/run_one_opts.py synth:femperm:110:110:110:3:random 16 4 --rm
PMC_SETS=$SHORT bash tools/pmc.sh gpurun_out/pmc_r05_holdout_n16    python $REPO/tools/run_one_opts.py synth:kron:850:sym:natural 16 4
PMC_SETS=$SHORT bash tools/pmc.sh gpurun_out/pmc_r05_holdout_n16_rm python $REPO/tools/run_one_opts.py synth:kron:850:sym:natural 16 4 --rm
for d in gpurun_out/pmc_r05_*; do echo "== $d"; grep -E "spmm_|repack|tiles_to" $d/summary.txt | awk '{print "  ", $1, $(NF-2), $(NF-1), $NF}' | cut -c1-200; done
