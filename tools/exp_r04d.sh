#!/bin/bash
# round-4: clustered plans of short-row matrices packed for a 320-row panel (20.5 KB LDS per workgroup) against the 576-row panel
cd "$(dirname "$0")/.."
for M in 160x160x160x1 synth:stencil2d:1400:1400:9:2 synth:stencil2d:2000:2000:9:1 synth:mesh3d:159:1:random; do
  for N in 16 128; do echo "## $M"; python tools/ab_opts.py $M $N 10 "small_panel=0" "small_panel=1"; done
done
