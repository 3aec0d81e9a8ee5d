#!/bin/bash
# PMC pass on the GPU box:  tools/pmc_run.sh <tag> "<COUNTER ...>" <python script> [args...]
# Writes gpurun_out/pmc/<tag>.csv = per-kernel mean of every counter (rocprofv3 --pmc with --kernel-trace only).
set -u
tag=$1; ctrs=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
timeout 900 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_$tag -o run -- python3 $R/"$1" "${@:2}" > $R/gpurun_out/pmc/${tag}.log 2>&1
f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
python3 - "$f" $R/gpurun_out/pmc/${tag}.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in agg.values() for c in k})
with open(sys.argv[2], "w") as f:
    f.write("kernel,dispatches," + ",".join(names) + "\n")
    for k, d in agg.items():
        n = max(len(v) for v in d.values())
        f.write(k.replace(",", ";") + f",{n}," + ",".join(f"{sum(d[c])/max(len(d[c]),1):.4g}" if c in d else "" for c in names) + "\n")
print(open(sys.argv[2]).read())
PY
rm -rf /tmp/pmc_$tag
