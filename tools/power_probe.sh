#!/bin/bash
# tools/power_probe.sh <workload> [engine options...] -- sample rocm-smi power / clocks while one workload loops
python $(pwd)/tools/run_one.py "$@" iters=${ITERS:-6000} > /tmp/pp_run.log 2>&1 &
PID=$!
for i in $(seq 1 24); do
  rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Package Power|sclk|GPU use" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 0.5
done
wait $PID
tail -1 /tmp/pp_run.log
