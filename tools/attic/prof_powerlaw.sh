#!/bin/bash
# per-kernel times of the strict-order power-law case: tools/prof_powerlaw.sh [run_powerlaw.py arguments]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_pl
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_pl -o pl -- python $GRAFT_REPO_ROOT/tools/run_powerlaw.py iters=10 "$@" 2>&1 | grep "powerlaw N"
find /tmp/rp_pl -name "*kernel_stats.csv" -exec python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'sx::' in r['Name'] and int(r['Calls']) > 2: print('   ', r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')" {} \;
