#!/bin/bash
# brick shapes with 32- / 64-row runs (C in whole 128-byte lines) on the short-row classes, same box (debug option cluster_shape)
export SEXTANS_DEBUG_OPTIONS=1
echo "== 27-point 1-dof 160^3"; python tools/ab_opts.py 160x160x160x1 16 20 cluster_shape=0 cluster_shape=320201 cluster_shape=320102 2>&1 | grep round
echo "== 9-point 2-D 2000^2 x 1"; python tools/ab_opts.py synth:stencil2d:2000:2000:9:1 16 20 cluster_shape=0 cluster_shape=320201 cluster_shape=640101 2>&1 | grep round
echo "== 9-point 2-D 1400^2 x 2"; python tools/ab_opts.py synth:stencil2d:1400:1400:9:2 16 20 cluster_shape=0 cluster_shape=320201 cluster_shape=640101 2>&1 | grep round
