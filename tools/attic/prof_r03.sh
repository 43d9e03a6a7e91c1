#!/bin/bash
# tools/prof_r03.sh -- the round-3 evidence set: kernel-trace stats of the bench, PMC passes (traffic) of the headline,
# PMC passes + power probe of the FEM panel kernel (v2), of FEM N=128 and of the block-banded MFMA kernel.
R=$(pwd)
bash tools/prof.sh r03
cd $R
unset PMC_SETS
bash tools/pmc.sh gpurun_out/pmc_r03_fem_n16 python $R/tools/run_one.py fem iters=5
bash tools/pmc.sh gpurun_out/pmc_r03_fem_n128 python $R/tools/run_one.py femN128 iters=3
export PMC_SETS="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY;GRBM_GUI_ACTIVE;TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum;FETCH_SIZE;WRITE_SIZE;SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
bash tools/pmc.sh gpurun_out/pmc_r03_bell_banded python $R/tools/run_bell.py 1048576 127 iters=3
unset PMC_SETS
ITERS=8000 bash tools/power_probe.sh fem > gpurun_out/power_r03_fem_n16.txt 2>&1
ITERS=2000 bash tools/power_probe.sh femN128 > gpurun_out/power_r03_fem_n128.txt 2>&1
(python $R/tools/run_bell.py 1048576 127 iters=4000 > gpurun_out/power_bell_run.log 2>&1 &)
sleep 12
for i in $(seq 1 16); do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Package Power|sclk|GPU use" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.5; done > gpurun_out/power_r03_bell_banded.txt
sleep 12
cat gpurun_out/power_bell_run.log >> gpurun_out/power_r03_bell_banded.txt
