R=$(pwd)
export PMC_SETS="TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum;SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY;GRBM_GUI_ACTIVE"
bash tools/pmc.sh gpurun_out/pmc_bell_shared2 python $R/tools/run_bell.py 1048576 64 bell_shared=1 iters=3
