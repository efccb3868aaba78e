set -o pipefail
cd /root/repo
for c in c2 c3 c4 c5; do
  python bench.py --config $c --steps 5 --warmup 1 > gpurun_out/r02_bench_$c.json 2> gpurun_out/r02_bench_$c.err || { echo "bench $c failed"; tail -5 gpurun_out/r02_bench_$c.err; }
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r02_bench_$c.json").read().strip().splitlines()[-1])
    r = d["roofline"]; cb = d.get("cpu_baseline", {})
    print("$c value %.3e ms/step %.1f kernel %s kernel_ms %.2f frac %.3f cpu %.3e (%s cores) 1thr %.3e" % (d["value"], d["ms_per_step"], r["kernel"], r["kernel_ms"], r["frac"], cb.get("value", 0), cb.get("cores"), cb.get("single_thread", {}).get("value", 0)))
except Exception as e:
    print("$c: no line", e)
PY
done
FMCMC_PERF_GUARD_RECORD=1 timeout -k 10 600 python -m pytest tests/test_gpu_perf_guard.py -x -q > gpurun_out/r02_perf_guard.log 2>&1; tail -3 gpurun_out/r02_perf_guard.log; cat gpurun_out/perf_guard.json
