#!/bin/bash
# per-kernel times of one sweep workload: tools/prof_sweep_one.sh <synth spec> <N> [--opt k=v ...]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_one
SPEC=$1; N=$2; shift 2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_one -o one -- python -m sextans_amd.sweep --rp 10 --n $N "$@" $SPEC 2>/dev/null | grep '^{' | cut -c1-260
find /tmp/rp_one -name "*kernel_stats.csv" -exec python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'sx::' in r['Name'] and int(r['Calls']) > 5: print('   ', r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')" {} \;
