#!/bin/bash
# Per-dispatch evidence for the byte ledger on the GPU box:  tools/ledger_run.sh <tag> [bench args...]
# Three rocprofv3 runs of the same short bench.py command, each on its own (counters never share a run with another trace domain):
#   1. --kernel-trace                 -> gpurun_out/ledger/<tag>_trace.csv.gz   (per dispatch: kernel, grid, workgroup, start / end)
#   2. --kernel-trace --pmc FETCH_SIZE -> gpurun_out/ledger/<tag>_fetch.csv.gz  (per dispatch counter rows)
#   3. --kernel-trace --pmc WRITE_SIZE -> gpurun_out/ledger/<tag>_write.csv.gz
#   4. --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES -> <tag>_mfma.csv.gz   (matrix-pipe busy cycles, summed over the chip's SIMDs)
#   5. --kernel-trace --pmc GRBM_GUI_ACTIVE          -> <tag>_gui.csv.gz    (GPU-active cycles, summed over the 8 XCDs)
# tools/ledger.py turns them into profiles/<tag>_ledger.{json,md}: class = kernel x grid size, bytes = (2 FETCH + WRITE) KiB,
# MFMA utilisation = MFMA busy cycles / (GPU-active cycles / 8 XCDs x 1024 SIMDs).
set -u
tag=${1:-run}; shift || true
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/ledger
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="--steps 2 --warmup 1 --no-cpu-baseline $*"
rm -rf /tmp/lg_$tag
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/lg_$tag/t -o run -- python3 $R/bench.py $args --gemm-seq $out/${tag}_gemm_seq.json > $out/${tag}_trace.log 2>&1 || exit 1
f=$(find /tmp/lg_$tag/t -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && gzip -c "$f" > $out/${tag}_trace.csv.gz
echo trace done
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/lg_$tag/f -o run -- python3 $R/bench.py $args > $out/${tag}_fetch.log 2>&1 || exit 1
f=$(find /tmp/lg_$tag/f -name '*counter_collection.csv' | head -1); [ -n "$f" ] && gzip -c "$f" > $out/${tag}_fetch.csv.gz
echo fetch done
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/lg_$tag/w -o run -- python3 $R/bench.py $args > $out/${tag}_write.log 2>&1 || exit 1
f=$(find /tmp/lg_$tag/w -name '*counter_collection.csv' | head -1); [ -n "$f" ] && gzip -c "$f" > $out/${tag}_write.csv.gz
echo write done
for pair in "SQ_VALU_MFMA_BUSY_CYCLES mfma" "GRBM_GUI_ACTIVE gui"; do
  set -- $pair
  timeout 600 rocprofv3 --kernel-trace --pmc $1 --output-format csv -d /tmp/lg_$tag/$2 -o run -- python3 $R/bench.py $args > $out/${tag}_$2.log 2>&1 || exit 1
  f=$(find /tmp/lg_$tag/$2 -name '*counter_collection.csv' | head -1); [ -n "$f" ] && gzip -c "$f" > $out/${tag}_$2.csv.gz
  echo $2 done
done
rm -rf /tmp/lg_$tag
ls -la $out
