#!/bin/bash
# small-panel kernel at 6 instead of 5 workgroups per CU (80 registers per lane): same-box A/B of two builds
L1=sextans_amd/lib/libsextans_amd.so; L2=sextans_amd/lib/libsextans_amd_w6.so
for spec in 160x160x160x1 synth:stencil2d:1400:1400:9:2 synth:stencil2d:2000:2000:9:1 synth:mesh3d:159:1:random; do
  echo "== $spec"; python tools/ab.py $spec 16,32 20 $L1 $L2
done
