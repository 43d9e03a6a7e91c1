#!/bin/bash
# round-4 same-box experiments: tile-group pipelining, exact vs FMA at wide N, brick shapes (C in full 128-byte lines)
export SEXTANS_DEBUG_OPTIONS=1
cd "$(dirname "$0")/.."
echo "## pipelining, FEM 3-dof 110^3"
for N in 32 128; do python tools/ab_opts.py 110x110x110x3 $N 10 "pipeline_tiles=0" "pipeline_tiles=1"; done
echo "## exact vs FMA"
for N in 64 128 256; do python tools/ab_opts.py 110x110x110x3 $N 6 "exact=1" "exact=0"; done
for N in 64 128; do python tools/ab_opts.py 160x160x160x1 $N 6 "exact=1" "exact=0"; done
for N in 64 128; do python tools/ab_opts.py synth:stencil2d:1400:1400:9:2 $N 6 "exact=1" "exact=0"; done
echo "## brick shapes (run_rows*10000 + lines*100 + planes), FEM 3-dof"
for N in 16 128; do python tools/ab_opts.py 110x110x110x3 $N 8 "cluster_shape=0" "cluster_shape=320201" "cluster_shape=320102" "cluster_shape=160401"; done
