#!/bin/bash
# Kernel-time profile of bench.py on the GPU box:  tools/profile_bench.sh <tag> [bench args...]
# Writes gpurun_out/prof/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats) and the bench JSON line next to it.
set -u
tag=${1:-run}; shift || true
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$tag -o bench -- \
    python3 $R/bench.py --no-cpu-baseline "$@" > $out/${tag}.log 2>&1
f=$(find /tmp/rp_$tag -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" $out/${tag}_kernel_stats.csv; fi
grep '^{"metric"' $out/${tag}.log > $out/${tag}_bench.json
rm -rf /tmp/rp_$tag
head -28 $out/${tag}_kernel_stats.csv | cut -c1-200
cat $out/${tag}_bench.json | cut -c1-200
