#!/bin/bash
# Round-5 evidence run on the GPU box (results under gpurun_out/final5/, copied into profiles/ by hand).  ORDER: the parity suite runs FIRST
# (a perf A/B without the parity tests is not a measurement); nothing below is kept if it fails.
#   1. pytest -m gpu
#   2. kernel-trace stats of `bench.py --eager` (one launch chain: the pass the roofline objects are sampled in) and of the default command
#      (eager pass + the graph-replayed micro-batch steps)
#   3. the five-pass ledger of `bench.py --eager` (trace, FETCH_SIZE, WRITE_SIZE, MFMA busy, GPU active: per dispatch)
#   4. the other workloads, the default line (with cpu_baseline) last
# One gpurun call is limited to 20 minutes: in practice the four parts were run as three calls (1; 2 + 3; 4), same order.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/final5
python -m pytest tests -m gpu -x -q > gpurun_out/final5/pytest_gpu.txt 2>&1 || { tail -5 gpurun_out/final5/pytest_gpu.txt; echo "PARITY SUITE FAILED: stopping"; exit 1; }
tail -2 gpurun_out/final5/pytest_gpu.txt
bash tools/quick_prof.sh r05_eager --eager > gpurun_out/final5/prof_eager.txt 2>&1
echo prof eager done
bash tools/quick_prof.sh r05_default > gpurun_out/final5/prof_default.txt 2>&1
echo prof default done
bash tools/ledger_run.sh r05_final --eager > gpurun_out/final5/ledger.txt 2>&1
echo ledger done
for w in vit_b swin_l avs avqa; do
  python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final5/bench_$w.json 2> gpurun_out/final5/bench_$w.err
  echo $w $(cut -c1-140 gpurun_out/final5/bench_$w.json)
done
python bench.py > gpurun_out/final5/bench_default.json 2> gpurun_out/final5/bench_default.err
cut -c1-300 gpurun_out/final5/bench_default.json
