set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/final
bash tools/profile_bench.sh r02_b32_v4 --steps 6 --warmup 2 --no-graph > gpurun_out/final/prof_v3.txt 2>&1
echo prof done
bash tools/pmc_run.sh r02_fetch "FETCH_SIZE" bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph > gpurun_out/final/pmc_fetch.txt 2>&1
echo fetch done
bash tools/pmc_run.sh r02_write "WRITE_SIZE" bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph > gpurun_out/final/pmc_write.txt 2>&1
echo write done
for w in vit_b swin_l avs avqa; do
  python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final/bench_$w.json 2> gpurun_out/final/bench_$w.err
  echo $w $(cut -c1-160 gpurun_out/final/bench_$w.json)
done
python bench.py --workload swin_l --fp8 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final/bench_swin_l_fp8.json 2> gpurun_out/final/bench_swin_l_fp8.err
python bench.py --workload avqa --fp8 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final/bench_avqa_fp8.json 2> gpurun_out/final/bench_avqa_fp8.err
echo fp8 done
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
cut -c1-300 gpurun_out/final/bench_default.json
