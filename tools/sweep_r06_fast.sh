#!/bin/bash
# tools/sweep_r06_fast.sh -- SEXTANS_MODE_FAST ("exact" 0 + "split_rows" -1; |d| <= 1e-4 * (|alpha| sum|a b| + |beta c|)) over the classes of the
# round-5 sweep, column-major and row-major entry points, next to one strict (default) pass of the same classes on the same box.
OUT=gpurun_out/r06_sweep_modes.jsonl
ERR=gpurun_out/r06_sweep_modes.err
: > $OUT; : > $ERR
NS=16,64,128,256
CLS="synth:fem3d:110:110:110:3 synth:fem3d:160:160:160:1 synth:stencil2d:1400:1400:9:2 synth:kkt:2000000:4 synth:femperm:110:110:110:3:random synth:mesh3d:110:3:random synth:kron:850:sym:natural synth:kron:850:sym:random synth:powerlaw:1000000:6:120:400000 synth:uniform:4000000:40"
for MODE in 0 1; do
  L=$([ $MODE = 1 ] && echo fast || echo strict)
  python -m sextans_amd.sweep --rp 20 --n $NS --opt mode=$MODE $CLS 2>>$ERR | grep '^{' | sed "s/\"matrix\": /\"mode\": \"$L\", \"matrix\": /" >> $OUT
  python -m sextans_amd.sweep --rp 20 --n $NS --opt mode=$MODE --rm $CLS 2>>$ERR | grep '^{' | sed "s/\"matrix\": /\"mode\": \"$L\", \"matrix\": /" >> $OUT
done
python -m sextans_amd.sweep --rp 50 --n 16,128 --opt mode=1 --check matrices/nasa4704/nasa4704.mtx tests/golden/cases/*.mtx 2>>$ERR | grep '^{' | sed 's/"matrix": /"mode": "fast", "matrix": /' >> $OUT
wc -l $OUT
