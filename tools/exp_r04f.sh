#!/bin/bash
# persistent small-panel launches (option persist) against the one-workgroup-per-block launch, same box
for spec in 160x160x160x1 synth:stencil2d:1400:1400:9:2 synth:stencil2d:2000:2000:9:1 synth:mesh3d:159:1:random synth:mesh3d:159:1:sweep; do
  for N in 16 32; do echo "== $spec N=$N"; python tools/ab_opts.py $spec $N 20 persist=0 persist=1 2>&1 | grep round; done
done
