#!/bin/bash
# 128-row bricks with 32-row runs (both halves of a 128-byte C line written by one workgroup) against the default 16 x 4 x 2 / 16 x 8, two row sets
export SEXTANS_DEBUG_OPTIONS=1
echo "== 27-point 1-dof 160^3"; for N in 16 128; do python tools/ab_opts.py 160x160x160x1 $N 20 row_sets=2 row_sets=3,cluster_shape=320202 row_sets=3,cluster_shape=320401 2>&1 | grep "round [12]"; done
echo "== 9-point 2-D 2000^2 x 1"; python tools/ab_opts.py synth:stencil2d:2000:2000:9:1 16 20 row_sets=1 row_sets=3 row_sets=3,cluster_shape=320401 row_sets=3,cluster_shape=640201 2>&1 | grep "round [12]"
