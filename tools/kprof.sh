#!/bin/bash
# per-kernel times of one tool run:  tools/kprof.sh <tag> <python script> [args...]   -> prints the kernel-stats rows (name, calls, avg us)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kp_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp_$tag -o k -- python3 "$@" > /tmp/kp_$tag.log 2>&1
f=$(find /tmp/kp_$tag -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f"{r['Name'][:100]:100s} calls={int(r['Calls']):5d} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
rm -rf /tmp/kp_$tag
