#!/bin/bash
# tools/power_probe.sh <workload> [engine options...] -- sample rocm-smi power / clocks while one workload loops
python tools/run_one.py "$@" iters=3000 > /tmp/pp_run.log 2>&1 &
PID=$!
sleep 6
for i in 1 2 3 4 5; do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|GPU use|fclk" | tr '\n' ';'; echo; sleep 1; done
wait $PID
tail -1 /tmp/pp_run.log
