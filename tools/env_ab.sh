#!/bin/bash
# Same-box A/B of one environment switch on the default bench line: tools/env_ab.sh NAME "v1 v2 ..." [rounds]
# prints the replayed (default-form) and eager ms per step of every run.
NAME=$1; VALS=$2; ROUNDS=${3:-2}
for r in $(seq 1 $ROUNDS); do
  for v in $VALS; do
    env $NAME=$v python bench.py --no-cpu-baseline --steps 10 --warmup 3 > /tmp/ab_$v.json 2>/tmp/ab_$v.err || { tail -5 /tmp/ab_$v.err; exit 1; }
    python - "$NAME" "$v" "$r" <<'PY'
import json, sys
d = json.loads(open(f"/tmp/ab_{sys.argv[2]}.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]}={sys.argv[2]} round {sys.argv[3]}: default form {d['ms_per_step']:.3f} ms ({d['value']:.1f} clips/s, {d['config'].get('step')}), eager pass {d['roofline_pass']['ms_per_step']:.3f} ms", flush=True)
PY
  done
done
