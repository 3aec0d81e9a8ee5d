#!/bin/bash
# Round-6 evidence run on the GPU box, in parts (one gpurun call is limited to 20 minutes); results under gpurun_out/final6/, copied into profiles/
# by hand.  ORDER: the parity suite first (a perf number without the parity tests is not a measurement).
#   tools/final_profiles_r06.sh tests       pytest -m gpu
#   tools/final_profiles_r06.sh prof        rocprofv3 --kernel-trace --stats of `bench.py --eager` and of the default command; the five-pass ledger
#                                           (trace, FETCH_SIZE, WRITE_SIZE, MFMA busy, GPU active) of `bench.py --eager` + tools/ledger.py
#   tools/final_profiles_r06.sh bench       the other workloads, the driver's command, the default line (with cpu_baseline) last + floor table
# LEDGER_COMMIT (set by the caller: the GPU box has no .git) is recorded in the PMC file.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
out=gpurun_out/final6
mkdir -p $out
case "${1:-tests}" in
tests)
  python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1 || { tail -5 $out/pytest_gpu.txt; echo "PARITY SUITE FAILED"; exit 1; }
  tail -2 $out/pytest_gpu.txt ;;
prof)
  bash tools/quick_prof.sh r06_eager --eager > $out/prof_eager.txt 2>&1; echo prof eager done
  bash tools/quick_prof.sh r06_default > $out/prof_default.txt 2>&1; echo prof default done
  bash tools/ledger_run.sh r06_final --eager > $out/ledger.txt 2>&1; echo ledger runs done
  export LEDGER_COMMAND="bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline"
  python tools/ledger.py gpurun_out/ledger/r06_final $out/r06_final gpurun_out/ledger/r06_final_gemm_seq.json > $out/ledger_py.txt 2>&1; tail -3 $out/ledger_py.txt ;;
bench)
  for w in vit_b swin_l avs avqa; do
    python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_$w.json 2> $out/bench_$w.err
    cp gpurun_out/bench_detail.json $out/detail_$w.json
    echo $w $(tail -1 $out/bench_$w.json | cut -c1-140)
  done
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_driver.json 2> $out/bench_driver.err
  echo driver $(tail -1 $out/bench_driver.json | cut -c1-200)
  python bench.py > $out/bench_default.json 2> $out/bench_default.err
  cp gpurun_out/bench_detail.json $out/detail_default.json
  python tools/floor_table.py $out/detail_default.json > $out/r06_floor.md
  tail -1 $out/bench_default.json | cut -c1-300 ;;
esac
