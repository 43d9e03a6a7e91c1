#!/bin/bash
# tools/evidence_r06.sh -- the round-6 evidence run on one MI355X box (run from the repo root through gpurun).
#  1. the bench line as the driver runs it (two stdout lines since round 6: {"also": ...} then the headline object) + the forced
#     1-rank distributed forms (sextans_dist_prepare + sextans_dist_spmm / _rm over a real 1-rank RCCL communicator);
#  2. kernel-trace statistics PER WORKLOAD (one rocprofv3 pass for the headline and one per `also` entry);
#  3. PMC passes of the headline (tools/prof.sh r06 -> config-4 traffic).
cd "$(dirname "$0")/.."
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_kernel_stats
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_stdout.txt 2> gpurun_out/r06_bench_stderr.txt
tail -1 gpurun_out/r06_bench_stdout.txt > gpurun_out/r06_bench_final.json
SEXTANS_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-also --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06_bench_forced_dist_colmajor.json
SEXTANS_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-also --no-cpu-baseline --rowmajor 2>/dev/null | tail -1 > gpurun_out/r06_bench_forced_dist_rowmajor.json
stats() {   # stats <key> <command...>
  local key=$1; shift
  rm -rf /tmp/rp_$key
  (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$key -o s -- "$@" > $REPO/gpurun_out/r06_kernel_stats/$key.log 2>&1)
  find /tmp/rp_$key -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06_kernel_stats/$key.csv \;
}
stats headline_config4 python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also
for KEY in $(python bench.py --only list 2>/dev/null | tail -1); do stats $KEY python $REPO/bench.py --only $KEY; done
python tools/merge_kernel_stats.py gpurun_out/r06_kernel_stats > gpurun_out/r06_bench_kernel_stats_per_workload.csv
bash tools/prof.sh r06
