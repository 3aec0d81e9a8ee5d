#!/bin/bash
# where do the 1.6x algorithmic bytes of the multi-tile 8-phase GEMM classes come from, and do they cost time?
# Per class (qkv: 125440 x 1536 x 512 + bias; fc1: x 2048 x 512 + bias + GELU + byte derivative) and routing (default multi-tile walk /
# STG_GEMM_8PHM=0 one tile per workgroup): separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE + L2 hit / miss | fabric read requests +
# their summed occupancy = latency) and one counter-free kernel trace for the duration.  -> gpurun_out/pmc/traffic_*.csv, summary on stdout.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
for cls in "125440 1536 512 bias" "125440 2048 512 gelu8" "62720 1536 512 bias" "62720 2048 512 gelu8"; do
  for m8 in 1 0; do
    tag=traffic_$(echo $cls | tr ' ' '_')_m$m8
    export STG_GEMM_8PHM=$m8
    bash $R/tools/pmc_run.sh ${tag}_p1 "FETCH_SIZE" tools/gemm_class.py $cls > /dev/null 2>&1
    bash $R/tools/pmc_run.sh ${tag}_p2 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" tools/gemm_class.py $cls > /dev/null 2>&1
    bash $R/tools/pmc_run.sh ${tag}_p3 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_REQ_sum" tools/gemm_class.py $cls > /dev/null 2>&1
    bash $R/tools/pmc_run.sh ${tag}_p4 "GRBM_GUI_ACTIVE" tools/gemm_class.py $cls > /dev/null 2>&1
    cd /tmp && rm -rf /tmp/tr_$tag && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$tag -o t -- python3 $R/tools/gemm_class.py $cls > /dev/null 2>&1
    f=$(find /tmp/tr_$tag -name '*kernel_stats.csv' | head -1); grep "gemm_nt" "$f" | head -2 | sed "s/^/$tag time: /"; rm -rf /tmp/tr_$tag; cd $R
    for ps in p1 p2 p3 p4; do grep -h "gemm_nt" $R/gpurun_out/pmc/${tag}_$ps.csv | sed "s/^/$tag $ps: /"; head -1 $R/gpurun_out/pmc/${tag}_$ps.csv | sed "s/^/$tag $ps hdr: /"; done
  done
done
