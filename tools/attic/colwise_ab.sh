#!/bin/bash
# tools/colwise_ab.sh -- same-box A/B of the lane-per-row kernel's tile placement (option "colwise_tiles_adjacent": 0 = tile as the slow
# grid axis, the form of rounds 3-5a; 1 = column-major: tiles of a row block neighbours in the launch order up to 4 tiles, row-major:
# groups of T lanes per row, one tile each; 2 = column-major: always neighbours), 4M-row 5-point stencil, both layouts.
for adj in 1 0 2; do
 for layout in "" "--rm"; do
  python -m sextans_amd.sweep --rp 20 --n 16,32,48,64,128,256 $layout --opt colwise_tiles_adjacent=$adj synth:stencil2d:2000:2000:5:1 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('colwise_tiles_adjacent=$adj layout=${layout:-cm}', r['N'], r['kernel'], r['ms'], r['roofline_frac'])
"
 done
done
