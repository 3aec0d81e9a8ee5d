#!/bin/bash
# Kernel-time profile of a short bench.py run on the GPU box:  tools/quick_prof.sh <tag> [bench args...]
# -> gpurun_out/prof/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_bench.json; prints the top kernels.
set -u
tag=${1:-run}; shift || true
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_$tag
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$tag -o bench -- \
    python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 2 "$@" > $out/${tag}.log 2>&1
f=$(find /tmp/rp_$tag -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $out/${tag}_kernel_stats.csv
grep '^{"metric"' $out/${tag}.log > $out/${tag}_bench.json
rm -rf /tmp/rp_$tag
python3 - $out/${tag}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 6.0
for r in rows[:40]:
    print(f"{r['Name'][:96]:96s} n/step={int(r['Calls'])/steps:7.1f} ms/step={float(r['TotalDurationNs'])/1e6/steps:7.3f} avg_us={float(r['AverageNs'])/1e3:8.1f}")
print("kernel ms/step:", round(sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps, 2))
PY
cut -c1-200 $out/${tag}_bench.json
