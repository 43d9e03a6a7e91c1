#!/bin/bash
# tools/prof.sh <tag> -- rocprofv3 kernel-trace stats + PMC passes of `python bench.py` on the GPU box.
# Writes summaries to gpurun_out/prof_<tag>/ (copy what should be judged into profiles/).
TAG=${1:-r01}; shift
EXTRA="$@"
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also $EXTRA"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -o stats -- $CMD > $OUT/bench_under_rocprof.log 2>&1
find /tmp/rp_stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout -k 10 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/rp_$N -o pmc -- $CMD > $OUT/pmc_$N.log 2>&1
  F=$(find /tmp/rp_$N -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then python $REPO/tools/pmc_summary.py $F > $OUT/pmc_$N.txt 2>&1; fi
done
rocprofv3 -L 2>/dev/null | grep -oE "(TCC|TCP|MALL|SQ_LDS|GRBM)[A-Za-z0-9_]*" | sort -u > $OUT/counters_available.txt
cd $REPO
