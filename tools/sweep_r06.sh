#!/bin/bash
# tools/sweep_r06.sh -- the paper-style sweep (reference README.md:18,31: N up to 512 over many matrices) on one MI355X: every synthetic
# class of round 4 + the HOLDOUT classes of round 6 (same classes as round 5) (kron(T_850, nasa4704) in three numberings, rectangular, unsymmetric pattern) at
# N in {8 ... 512}, column-major entry points; every class once more through the ROW-major entry point
# (records carry "layout": "rm").  One JSON record per (matrix, N); step time = layout passes + kernels.
OUT=gpurun_out/r06_sweep.jsonl
ERR=gpurun_out/r06_sweep.err
: > $OUT; : > $ERR
NS=8,16,32,64,128,256,512
HOLD="synth:kron:850:sym:natural synth:kron:850:sym:random synth:kron:850:sym:rcm synth:kron:850:rect:natural synth:kron:850:unsym:natural synth:kron:850:unsym:random"
python -m sextans_amd.sweep --rp 20 --n $NS \
  synth:uniform:4000000:40 synth:banded:4000000:40:2000 synth:fem3d:110:110:110:3 synth:fem3d:160:160:160:1 \
  synth:stencil2d:2000:2000:5:1 synth:stencil2d:1400:1400:9:2 synth:kkt:2000000:4 \
  synth:femperm:110:110:110:3:random synth:femperm:110:110:110:3:rcm synth:mesh3d:110:3:sweep synth:mesh3d:110:3:random \
  synth:mesh3d:159:1:random $HOLD 2>>$ERR | grep '^{' >> $OUT
python -m sextans_amd.sweep --rp 20 --n $NS --rm synth:uniform:4000000:40 synth:banded:4000000:40:2000 synth:stencil2d:2000:2000:5:1 synth:kkt:2000000:4 \
  synth:fem3d:110:110:110:3 synth:fem3d:160:160:160:1 synth:stencil2d:1400:1400:9:2 \
  synth:femperm:110:110:110:3:random synth:mesh3d:110:3:random synth:mesh3d:159:1:random $HOLD 2>>$ERR | grep '^{' >> $OUT
python -m sextans_amd.sweep --rp 20 --n $NS --opt row_cluster=0 synth:femperm:110:110:110:3:random 2>>$ERR | grep '^{' | sed 's/"matrix": "synth:femperm/"options": "row_cluster=0 (natural-order forms)", "matrix": "synth:femperm/' >> $OUT
python -m sextans_amd.sweep --rp 20 --n 64,128,256 --opt exact=0 synth:fem3d:110:110:110:3 2>>$ERR | grep '^{' | sed 's/"matrix": "synth:fem3d/"options": "exact=0 (FMA, opt-in)", "matrix": "synth:fem3d/' >> $OUT
python -m sextans_amd.sweep --rp 20 --n $NS --opt split_rows=-1 synth:powerlaw:1000000:6:120:400000 2>>$ERR | grep '^{' | sed 's/"matrix": "synth:powerlaw/"options": "split_rows=-1", "matrix": "synth:powerlaw/' >> $OUT
python -m sextans_amd.sweep --rp 20 --n $NS synth:powerlaw:1000000:6:120:400000 2>>$ERR | grep '^{' | sed 's/"matrix": "synth:powerlaw/"options": "default (strict order)", "matrix": "synth:powerlaw/' >> $OUT
python -m sextans_amd.sweep --rp 20 --n $NS --rm synth:powerlaw:1000000:6:120:400000 2>>$ERR | grep '^{' | sed 's/"matrix": "synth:powerlaw/"options": "default (strict order)", "matrix": "synth:powerlaw/' >> $OUT
python -m sextans_amd.sweep --rp 50 --n $NS --check matrices/nasa4704/nasa4704.mtx tests/golden/cases/*.mtx 2>>$ERR | grep '^{' >> $OUT
wc -l $OUT
