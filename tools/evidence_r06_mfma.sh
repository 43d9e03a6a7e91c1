#!/bin/bash
# tools/evidence_r06_mfma.sh -- VERDICT r05 task 6: rocprofv3 --pmc evidence for the bf16 MFMA path AT HEAD.
#   separate counter passes (tools/pmc.sh) of `bench.py --only config5_blocked_ell_bf16_N256` and `--only blockbanded_ell_bf16_N256`
#   -> gpurun_out/r06_mfma/{config5,banded}/summary.txt  (copy to profiles/r06_config5_pmc.txt, profiles/r06_blockbanded_pmc.txt)
cd "$(dirname "$0")/.."
REPO=$(pwd)
export PMC_SETS="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY;GRBM_GUI_ACTIVE;TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum;FETCH_SIZE;WRITE_SIZE"
export PMC_PASS_TIMEOUT=600
for w in config5:config5_blocked_ell_bf16_N256 banded:blockbanded_ell_bf16_N256; do
  d=${w%%:*}; key=${w##*:}
  bash tools/pmc.sh gpurun_out/r06_mfma/$d python $REPO/bench.py --only $key
  grep -E "spmm_bell|counter" gpurun_out/r06_mfma/$d/summary.txt > gpurun_out/r06_mfma/$d/bell.txt
done
