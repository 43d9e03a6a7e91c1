#!/bin/bash
# panel threshold experiment: 2-D 5-point stencil (reuse ~1.65 per 64-row block) on the gather kernel vs the panel kernel
for T in 200 150 120; do
  echo "== panel_min_reuse_x100=$T"
  python -m sextans_amd.sweep --rp 5 --n 8,16,32,128 --opt panel_min_reuse_x100=$T synth:stencil2d:2000:2000:5:1 synth:stencil2d:1400:1400:5:2 synth:banded:4000000:40:200 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['matrix'], r['N'], r['kernel'], r['ms'], r['roofline_frac'])"
done
