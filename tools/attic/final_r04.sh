#!/bin/bash
# tools/final_r04.sh -- the measurements of the final HEAD that are not in tools/evidence_r04.sh: bench line, sweep, renumbered classes, plan figures
cd "$(dirname "$0")/../.."
python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err
tail -c 600 gpurun_out/r04_bench_final.json
bash tools/sweep_r04.sh
{ python tools/reorder_big.py 110 3 16,128 natural,random,rcm,mesh_sweep,mesh_random; python tools/reorder_big.py 159 1 16 mesh_sweep,mesh_random; } 2>/dev/null | sed 's/"matrix": "mesh_\(sweep\|random\)", "M": 4019679/"matrix": "mesh1dof_\1", "M": 4019679/' > gpurun_out/r04_renumbered_classes.jsonl
python tools/plan_stats.py 2>/dev/null > gpurun_out/r04_plan_stats.txt
cat gpurun_out/r04_plan_stats.txt
