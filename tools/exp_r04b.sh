#!/bin/bash
# round-4 same-box experiments: panels staged straight from column-major B for LARGE matrices (fuse_b = 2: no repack launch)
export SEXTANS_DEBUG_OPTIONS=1
cd "$(dirname "$0")/.."
for M in 110x110x110x3 160x160x160x1 synth:stencil2d:1400:1400:9:2; do
  echo "## $M N=16"
  python tools/ab_opts.py $M 16 10 "fuse_b=1" "fuse_b=2"
done
echo "## 5-point N=16 (panel threshold lowered to 1.5 so that the panel kernel runs at all)"
python tools/ab_opts.py synth:stencil2d:2000:2000:5:1 16 10 "fuse_b=1" "fuse_b=1,panel_min_reuse_x100=150" "fuse_b=2,panel_min_reuse_x100=150" "row_cluster=2,panel_min_reuse_x100=150"
