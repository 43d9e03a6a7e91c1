#!/bin/bash
# tools/pmc.sh <outdir> <cmd...> : several rocprofv3 --pmc passes of one command, summarised per kernel
REPO=$(pwd)
OUT=$REPO/$1; shift
mkdir -p $OUT; rm -f $OUT/summary.txt
export TMPDIR=/tmp
i=0
IFS=";" read -ra SETS <<< "${PMC_SETS:-}"
if [ ${#SETS[@]} -eq 0 ]; then SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"); fi
for C in "${SETS[@]}"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 ${PMC_PASS_TIMEOUT:-420} rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$$_$i -o p -- "$@" > $OUT/pass$i.log 2>&1)
  F=$(find /tmp/pmc_$$_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then python $REPO/tools/pmc_summary.py $F | grep -E "spmm_|repack|counter" >> $OUT/summary.txt; else echo "pass $i ($C): no output" >> $OUT/summary.txt; tail -3 $OUT/pass$i.log >> $OUT/summary.txt; fi
done
