# tools/gpu_round_evidence.sh <tag>: the end-of-round evidence, one gpurun call: perf-guard record, the default bench line
# (headline + the other configs + cpu_baseline), the full `--config` lines, rocprofv3 stats + PMC passes of all four configs.
# Afterwards: tools/collect_profiles.sh <tag>_c2 <tag>_c3 <tag>_c4 <tag>_c5 ; cp gpurun_out/<tag>_bench_*.json gpurun_out/perf_guard.json profiles/
set -e
TAG=${1:-r03_b}
mkdir -p gpurun_out
rm -f gpurun_out/perf_guard.json
FMCMC_PERF_GUARD_RECORD=1 timeout -k 10 600 python -m pytest tests/test_gpu_perf_guard.py -q -m gpu > gpurun_out/perf_guard_record.log 2>&1 || { tail -20 gpurun_out/perf_guard_record.log; exit 1; }
cat gpurun_out/perf_guard.json
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err || { tail -5 gpurun_out/${TAG}_bench_default.err; exit 1; }
tail -c 900 gpurun_out/${TAG}_bench_default.json; echo
for c in c3 c4 c5; do
  timeout -k 10 600 python bench.py --config $c --steps 5 --warmup 1 > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err || { tail -5 gpurun_out/${TAG}_bench_$c.err; exit 1; }
  tail -c 500 gpurun_out/${TAG}_bench_$c.json; echo
done
for c in c2 c3 c4 c5; do
  timeout -k 10 900 bash tools/profile_bench.sh ${TAG}_$c $c full > gpurun_out/prof_${TAG}_$c.log 2>&1 || true
  cat gpurun_out/prof_${TAG}_$c/passes.log | tr '\n' ' '; echo
done
echo evidence done
