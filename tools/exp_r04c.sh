#!/bin/bash
# round-4: the lane-per-row kernel (kernel = 4) against the automatic choice on short- and medium-row classes
cd "$(dirname "$0")/.."
for M in synth:stencil2d:2000:2000:5:1 synth:stencil2d:2000:2000:9:1 synth:stencil2d:1400:1400:9:2 160x160x160x1 synth:banded:4000000:10:2000 synth:banded:4000000:40:2000; do
  for N in 16 32 128; do echo "## $M"; python tools/ab_opts.py $M $N 8 "colwise_max_len=0" "kernel=4"; done
done
