#!/bin/bash
# SQ counters of the 8-phase GEMM on 125440 x 2048 x 512 (and the long-K shape), with and without the epilogue (diagnostics build):
#   tools/gemm_pmc.sh          -> gpurun_out/pmc/gemm8ph_*.csv
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
export STGCMA_LIB=$R/stg-cma_amd/libstgcma_hip_diag.so STG_GEMM_8PH=2
for dbg in 0 3; do
  for shape in "125440 2048 512" "125440 512 2048"; do
    tag=gemm8ph_$(echo $shape | tr ' ' 'x')_dbg$dbg
    STG_GEMM_DBG=$dbg bash $R/tools/pmc_run.sh ${tag}_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" tools/gemm_one.py $shape > /dev/null 2>&1
    STG_GEMM_DBG=$dbg bash $R/tools/pmc_run.sh ${tag}_b "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" tools/gemm_one.py $shape > /dev/null 2>&1
    grep -h "8ph" $R/gpurun_out/pmc/${tag}_a.csv $R/gpurun_out/pmc/${tag}_b.csv | sed "s/^/$tag: /"
    head -1 $R/gpurun_out/pmc/${tag}_a.csv | sed "s/^/$tag hdr_a: /"; head -1 $R/gpurun_out/pmc/${tag}_b.csv | sed "s/^/$tag hdr_b: /"
  done
done
