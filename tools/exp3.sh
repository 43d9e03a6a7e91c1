R=$(pwd)
bash tools/pmc.sh gpurun_out/pmc_femN128_wide python $R/tools/run_one.py femN128 cols_per_lane=8 iters=3
bash tools/pmc.sh gpurun_out/pmc_femN128_c4 python $R/tools/run_one.py femN128 cols_per_lane=4 iters=3
