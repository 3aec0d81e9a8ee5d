#!/bin/bash
# SQ counters of the mha.hip kernels on the Swin-L stage-0 cross-modal shape and the ViT-B shape -> gpurun_out/pmc/mha_*.csv
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
for kind in xl vit; do
  python3 $R/tools/mha_one.py $kind 5
  bash $R/tools/pmc_run.sh mha_${kind}_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" tools/mha_one.py $kind 2 > /dev/null 2>&1
  bash $R/tools/pmc_run.sh mha_${kind}_b "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE" tools/mha_one.py $kind 2 > /dev/null 2>&1
  for ps in a b; do head -1 $R/gpurun_out/pmc/mha_${kind}_$ps.csv; grep -h "mha_" $R/gpurun_out/pmc/mha_${kind}_$ps.csv; done
done
