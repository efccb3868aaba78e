set -e
rm -f gpurun_out/perf_guard.json
FMCMC_PERF_GUARD_RECORD=1 timeout -k 10 600 python -m pytest tests/test_gpu_perf_guard.py -q -m gpu > gpurun_out/perf_guard_record.log 2>&1 || { tail -20 gpurun_out/perf_guard_record.log; exit 1; }
cat gpurun_out/perf_guard.json
for c in c2 c3 c4 c5; do
  timeout -k 10 600 python bench.py --config $c --steps 5 --warmup 1 > gpurun_out/r02_c_bench_$c.json 2> gpurun_out/r02_c_bench_$c.err || { tail -5 gpurun_out/r02_c_bench_$c.err; exit 1; }
  tail -c 600 gpurun_out/r02_c_bench_$c.json; echo
done
timeout -k 10 900 bash tools/profile_bench.sh r02_c_c2 c2 full > gpurun_out/prof_r02_c_c2.log 2>&1
timeout -k 10 900 bash tools/profile_bench.sh r02_c_c4 c4 full > gpurun_out/prof_r02_c_c4.log 2>&1
timeout -k 10 600 bash tools/profile_bench.sh r02_c_c3 c3 stats > gpurun_out/prof_r02_c_c3.log 2>&1
timeout -k 10 600 bash tools/profile_bench.sh r02_c_c5 c5 stats > gpurun_out/prof_r02_c_c5.log 2>&1
echo profiles done
