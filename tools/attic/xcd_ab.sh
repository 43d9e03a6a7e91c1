#!/bin/bash
# tools/xcd_ab.sh <tag> -- VERDICT r04 task 3: contiguous vs round-robin XCD placement of the reordered form at N = 16 / 32 on ONE box:
# kernel time over >= 200 launches each (tools/ab_opts.py, round-robin between the settings, three rounds) + HBM traffic of the
# kernel under each setting (separate rocprofv3 --pmc passes).  Run it on >= 3 boxes; the records go to gpurun_out/xcd_ab_<tag>.txt.
export SEXTANS_DEBUG_OPTIONS=1
OUT=gpurun_out/xcd_ab_$1.txt
: > $OUT
for SPEC in synth:femperm:110:110:110:3:random synth:mesh3d:110:3:random synth:kron:850:sym:random; do
  for N in 16 32; do
    echo "== $SPEC N=$N (kernel us, 200 launches per setting and round)" >> $OUT
    python tools/ab_opts.py $SPEC $N 200 reordered_xcd=1 reordered_xcd=0 2>&1 | grep "round" >> $OUT
  done
done
for X in 0 1; do
  for N in 16 32; do
    PMC_SETS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" PMC_PASS_TIMEOUT=300 bash tools/pmc.sh gpurun_out/xcd_pmc_$1_x${X}_n$N python $(pwd)/tools/run_one_opts.py synth:femperm:110:110:110:3:random $N 6 reordered_xcd=$X
    echo "== femperm random N=$N reordered_xcd=$X: counters per launch of the SpMM kernel (FETCH_SIZE / WRITE_SIZE in KB; reads = 2 x FETCH_SIZE on gfx950)" >> $OUT
    grep "spmm_csr_panel_v2" gpurun_out/xcd_pmc_$1_x${X}_n$N/summary.txt | awk '{print "   ", $(NF-2), $(NF-1), $NF}' >> $OUT
  done
done
cat $OUT
