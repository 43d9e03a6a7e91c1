#!/bin/bash
# tools/tpw_ab.sh -- same-box A/B of "tiles_per_wg" (super tiles one workgroup of spmm_csr_panel_v2 walks; 0 = the dispatcher's rule) on
# short-row and long-row classes, row-major and column-major entry points.
for m in "$@"; do
for layout in "--rm" ""; do
for tpw in 0 1 2; do
  python -m sextans_amd.sweep --rp 20 --n 32,64,128,256 $layout --opt tiles_per_wg=$tpw $m 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('$m tiles_per_wg=$tpw ${layout:-cm}', r['N'], r['kernel'], r['ms'], r['roofline_frac'])
"
done; done; done
