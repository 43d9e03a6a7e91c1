#!/bin/bash
# tools/exp.sh <experiment> -- the same-box experiments behind profiles/r0*_experiment_*.txt, one parameterised script (round 5: the ten
# tools/exp_r04a..j.sh folded into this).  Every experiment is a round-robin A/B on ONE box: option sets (tools/ab_opts.py) or
# library builds (tools/ab.py; variant builds: tools/build_variant.py <name> -DMACRO=...).  `tools/exp.sh list` names them.
export SEXTANS_DEBUG_OPTIONS=1
cd "$(dirname "$0")/.."
FEM=110x110x110x3; FEM1=160x160x160x1; S9x2=synth:stencil2d:1400:1400:9:2; S9=synth:stencil2d:2000:2000:9:1; S5=synth:stencil2d:2000:2000:5:1
SHORT="$FEM1 $S9x2 $S9 synth:mesh3d:159:1:random"
opts() { local spec=$1 n=$2 it=$3; shift 3; echo "== $spec N=$n"; python tools/ab_opts.py $spec $n $it "$@" 2>&1 | grep round; }
case "$1" in
  pipelining)      for N in 32 128; do opts $FEM $N 10 pipeline_tiles=0 pipeline_tiles=1; done ;;                                   # r04a
  fma)             for N in 64 128 256; do opts $FEM $N 6 exact=1 exact=0; done
                   for M in $FEM1 $S9x2; do for N in 64 128; do opts $M $N 6 exact=1 exact=0; done; done ;;                       # r04a
  brick_shapes)    for N in 16 128; do opts $FEM $N 8 cluster_shape=0 cluster_shape=320201 cluster_shape=320102 cluster_shape=160401; done   # r04a, r04h
                   opts $FEM1 16 20 cluster_shape=0 cluster_shape=320201 cluster_shape=320102
                   for M in $S9 $S9x2; do opts $M 16 20 cluster_shape=0 cluster_shape=320201 cluster_shape=640101; done ;;
  colmajor_large_b) for M in $FEM $FEM1 $S9x2; do opts $M 16 10 fuse_b=1 fuse_b=2; done                                              # r04b
                   opts $S5 16 10 fuse_b=1 fuse_b=1,panel_min_reuse_x100=150 fuse_b=2,panel_min_reuse_x100=150 row_cluster=2,panel_min_reuse_x100=150 ;;
  colwise)         for M in $S5 $S9 $S9x2 $FEM1 synth:banded:4000000:10:2000 synth:banded:4000000:40:2000; do                       # r04c
                     for N in 16 32 128; do opts $M $N 8 colwise_max_len=0 kernel=4; done; done ;;
  small_panel)     for M in $SHORT; do for N in 16 128; do opts $M $N 10 small_panel=0 small_panel=1; done; done ;;                # r04d
  row_sets)        for M in $FEM1 $S9x2 $S9 $S5; do for N in 16 32 128; do opts $M $N 20 row_sets=1 row_sets=2; done; done        # r04i, r04j
                   for N in 16 128; do opts $FEM1 $N 20 row_sets=2 row_sets=3,cluster_shape=320202 row_sets=3,cluster_shape=320401; done
                   opts $S9 16 20 row_sets=1 row_sets=3 row_sets=3,cluster_shape=320401 row_sets=3,cluster_shape=640201 ;;
  libs)            # tools/exp.sh libs "<spec> ..." <N list> <iters> lib1.so lib2.so ...   (r04e: 6 workgroups per CU; r04g: timing-only builds)
                   specs=$2; ns=$3; it=$4; shift 4
                   for M in $specs; do echo "== $M"; python tools/ab.py $M $ns $it "$@" 2>&1 | grep -v amdgpu.ids; done ;;
  xcd_placement)   shift; bash tools/xcd_ab.sh "${1:-box}" ;;                                                                       # r05, VERDICT r04 task 3
  spills)          python tools/ab.py synth:fem3d:40:40:40:1 16,128 300 sextans_amd/lib/libsextans_amd.so tools/bin/libsextans_bcol4.so 2>&1 | grep -v amdgpu.ids   # r05
                   for i in 1 2; do python tools/mixed_exp.py sextans_amd/lib/libsextans_amd.so 16 64; python tools/mixed_exp.py tools/bin/libsextans_mixed3.so 16 64; done 2>&1 | grep -v amdgpu.ids ;;
  list|*)          grep -oE "^  [a-z_]+\)" "$0" | tr -d ' )' ;;
esac
