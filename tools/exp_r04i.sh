#!/bin/bash
# two row sets per block (128-row bricks, option row_sets) against 64-row bricks on the short-row classes, same box
for spec in 160x160x160x1 synth:stencil2d:1400:1400:9:2 synth:stencil2d:2000:2000:9:1 synth:stencil2d:2000:2000:5:1; do
  for N in 16 32 128; do echo "== $spec N=$N"; python tools/ab_opts.py $spec $N 20 row_sets=1 row_sets=2 2>&1 | grep round; done
done
