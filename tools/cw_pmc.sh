#!/bin/bash
# SQ counters of conv_wgrad_kernel on one decoder shape (index $1 of tools/conv_wgrad_bench.py) -> gpurun_out/pmc/cw_*.csv
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
i=${1:-0}
bash $R/tools/pmc_run.sh cw_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" tools/conv_wgrad_bench.py 2 $i > /dev/null 2>&1
bash $R/tools/pmc_run.sh cw_b "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA" tools/conv_wgrad_bench.py 2 $i > /dev/null 2>&1
bash $R/tools/pmc_run.sh cw_c "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" tools/conv_wgrad_bench.py 2 $i > /dev/null 2>&1
for ps in a b c; do head -1 $R/gpurun_out/pmc/cw_$ps.csv; grep -h "conv_wgrad" $R/gpurun_out/pmc/cw_$ps.csv; done
