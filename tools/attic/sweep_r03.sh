#!/bin/bash
# tools/sweep_r03.sh -- the paper-style sweep (reference README.md:18,31: N up to 512 over many matrices) on one MI355X:
# every synthetic class + the .mtx fixtures at N in {8,16,32,64,128,256,512}.  One JSON record per (matrix, N).
OUT=gpurun_out/r03_sweep.jsonl
: > $OUT
NS=8,16,32,64,128,256,512
python -m sextans_amd.sweep --rp 5 --n $NS \
  synth:uniform:4000000:40 synth:banded:4000000:40:2000 synth:fem3d:110:110:110:3 synth:fem3d:160:160:160:1 \
  synth:stencil2d:2000:2000:5:1 synth:stencil2d:1400:1400:9:2 synth:kkt:2000000:4 2>>gpurun_out/r03_sweep.err | grep '^{' >> $OUT
python -m sextans_amd.sweep --rp 5 --n $NS --opt split_rows=-1 synth:powerlaw:1000000:6:120:400000 2>>gpurun_out/r03_sweep.err | grep '^{' | sed 's/"matrix": "synth:powerlaw/"options": "split_rows=-1", "matrix": "synth:powerlaw/' >> $OUT
python -m sextans_amd.sweep --rp 3 --n $NS synth:powerlaw:1000000:6:120:400000 2>>gpurun_out/r03_sweep.err | grep '^{' | sed 's/"matrix": "synth:powerlaw/"options": "default (strict order)", "matrix": "synth:powerlaw/' >> $OUT
python -m sextans_amd.sweep --rp 50 --n $NS --check matrices/nasa4704/nasa4704.mtx tests/golden/cases/*.mtx 2>>gpurun_out/r03_sweep.err | grep '^{' >> $OUT
wc -l $OUT
