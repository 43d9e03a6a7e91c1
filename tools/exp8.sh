#!/bin/bash
# TA / TD / TCP utilisation of the panel v2 kernel at FEM N=16 and N=128
R=$(pwd)
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TD|TCP)_[A-Za-z0-9_]*" | sort -u > gpurun_out/ta_counters.txt
export PMC_SETS="TA_TA_BUSY_sum TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE;TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum;TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum;TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum;TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum;TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
bash tools/pmc.sh gpurun_out/pmc_ta_fem_n128 python $R/tools/run_one.py femN128 iters=3
bash tools/pmc.sh gpurun_out/pmc_ta_fem_n16 python $R/tools/run_one.py fem iters=5
