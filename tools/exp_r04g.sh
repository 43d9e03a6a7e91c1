#!/bin/bash
# what bounds the short-row launches: timing-only builds (wrong results) that drop one class of memory accesses each
# v1: 1 of 4 C_in loads / C stores per lane; v2: only the first 64 dictionary rows of a panel; v3: no index-list loads; v4: no value loads
export SX_AB_OPTS=persist=0
L=sextans_amd/lib
for spec in 160x160x160x1 synth:stencil2d:2000:2000:9:1; do
  echo "== $spec"; python tools/ab.py $spec 16 20 $L/libsextans_amd.so $L/libsx_v1.so $L/libsx_v2.so $L/libsx_v3.so $L/libsx_v4.so 2>&1 | grep -v amdgpu.ids
done
