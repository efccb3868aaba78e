set -e
mkdir -p gpurun_out/r03
rocprofv3 -L > gpurun_out/r03/counters.txt 2>&1 || true
grep -c . gpurun_out/r03/counters.txt
timeout -k 10 600 bash tools/profile_bench.sh r03_a_c5 c5 full > gpurun_out/r03/prof_a_c5.log 2>&1
tail -60 gpurun_out/r03/prof_a_c5.log
cat gpurun_out/prof_r03_a_c5/passes.log
timeout -k 10 600 bash tools/profile_bench.sh r03_a_c3 c3 full > gpurun_out/r03/prof_a_c3.log 2>&1
tail -40 gpurun_out/r03/prof_a_c3.log
cat gpurun_out/prof_r03_a_c3/passes.log
timeout -k 10 300 python bench.py --steps 20 --warmup 2 > gpurun_out/r03/bench_default_a.json 2> gpurun_out/r03/bench_default_a.err || { tail -20 gpurun_out/r03/bench_default_a.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03/bench_default_a.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], {k:(v["value"],v["frac"],v["wall_s_incl_setup"]) for k,v in d["configs"].items()})
print(d["cpu_baseline"])
PY
