#!/bin/bash
# tools/evidence_r04.sh -- the round-4 evidence run on one MI355X box: kernel-trace stats of the full bench at HEAD, config-4 PMC
# passes (tools/prof.sh), PMC of the reordered form and of FEM N = 128, the per-rank slab times.  Summaries land in gpurun_out/.
cd "$(dirname "$0")/../.."
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_full -o full -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/r04_bench_under_rocprof.log 2>&1)
find /tmp/rp_full -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_bench_full_kernel_stats.csv \;
bash tools/prof.sh r04
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_reo -o reo -- python $REPO/tools/run_reordered.py 16 10 > $REPO/gpurun_out/r04_reordered_under_rocprof.log 2>&1)
find /tmp/rp_reo -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_reordered_n16_kernel_stats.csv \;
PMC_SETS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum;TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum;TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY;SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES;SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU;GRBM_GUI_ACTIVE" \
  bash tools/pmc.sh gpurun_out/pmc_r04_reordered_n16 python $REPO/tools/run_reordered.py 16 4
PMC_SETS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY;SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES;SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU;GRBM_GUI_ACTIVE" \
  bash tools/pmc.sh gpurun_out/pmc_r04_fem_n128 python $REPO/tools/run_one.py femN128 iters=3
PMC_SETS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum;GRBM_GUI_ACTIVE" \
  bash tools/pmc.sh gpurun_out/pmc_r04_fem_n16 python $REPO/tools/run_one.py fem iters=5
python tools/rank_slabs.py > gpurun_out/r04_rank_slab_times.json 2> gpurun_out/r04_rank_slab_times.log
tail -8 gpurun_out/r04_rank_slab_times.log
cat gpurun_out/pmc_r04_reordered_n16/summary.txt | head -60
