#!/bin/bash
# Round-3 evidence run on the GPU box (results under gpurun_out/final3/, copied into profiles/ by hand):
#   kernel-trace stats of the headline bench, the three-pass ledger (trace + FETCH_SIZE + WRITE_SIZE, per dispatch), the default bench line
#   (with cpu_baseline), the graph-replay line, the other workloads.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/final3
bash tools/quick_prof.sh r03_final > gpurun_out/final3/prof.txt 2>&1
echo prof done
bash tools/ledger_run.sh r03_final > gpurun_out/final3/ledger.txt 2>&1
echo ledger done
for w in vit_b swin_l avs avqa; do
  python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final3/bench_$w.json 2> gpurun_out/final3/bench_$w.err
  echo $w $(cut -c1-140 gpurun_out/final3/bench_$w.json)
done
python bench.py --graph --no-cpu-baseline > gpurun_out/final3/bench_graph.json 2> gpurun_out/final3/bench_graph.err
echo graph done
python bench.py > gpurun_out/final3/bench_default.json 2> gpurun_out/final3/bench_default.err
cut -c1-300 gpurun_out/final3/bench_default.json
