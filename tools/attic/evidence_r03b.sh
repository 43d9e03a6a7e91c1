#!/bin/bash
# second batch of round-3 evidence: full-bench kernel stats, strict power-law per kernel, microbenchmarks
R=$(pwd)
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/rp_full && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_full -o full -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_full_r03.log 2>&1; find /tmp/rp_full -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r03_bench_full_kernel_stats.csv \;)
{ for n in 16 32 128; do bash tools/prof_powerlaw.sh N=$n; done; } > gpurun_out/r03_powerlaw_strict_kernels.txt 2>&1
tools/bin/chain_bench > gpurun_out/r03_chain_bench.txt 2>&1
tools/bin/valu_bench > gpurun_out/r03_valu_bench.txt 2>&1
