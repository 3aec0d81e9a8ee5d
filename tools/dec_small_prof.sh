#!/bin/bash
# rocprofv3 kernel stats of tools/dec_small_bench.py -> gpurun_out/prof_dec/ ; prints the non-torch rows
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dec -o dec -- python3 $R/tools/dec_small_bench.py > $R/gpurun_out/dec_small.log 2>&1
cd $R
f=$(find gpurun_out/prof_dec -name '*kernel_stats.csv' | tail -1)
if [ -z "$f" ]; then echo "no stats file"; tail -5 gpurun_out/dec_small.log; exit 1; fi
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "at::" in r["Name"]: continue
    print(r["Name"].split("::")[-1].split("(")[0], r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "min_us", round(int(r["MinNs"]) / 1e3, 1), "max_us", round(int(r["MaxNs"]) / 1e3, 1))
PY
