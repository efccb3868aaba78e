set -e
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu -k "adapt or readme or session or continuation or G2 or G3 or gelman or not_pd or status" > gpurun_out/c3_tests.log 2>&1 || { tail -40 gpurun_out/c3_tests.log; exit 1; }
tail -2 gpurun_out/c3_tests.log
timeout -k 10 300 python bench.py --config c3 --steps 3 --warmup 1 > gpurun_out/c3_bench.json 2> gpurun_out/c3_bench.err
python - <<PY
import json; d=json.loads(open("gpurun_out/c3_bench.json").read().strip().splitlines()[-1]); print("c3", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
PY
FMCMC_AMD_DEBUG=mode=512 timeout -k 10 300 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/c3_bench_old.json 2> gpurun_out/c3_bench_old.err
python - <<PY
import json; d=json.loads(open("gpurun_out/c3_bench_old.json").read().strip().splitlines()[-1]); print("c3 (mode=512: single-outcome owner)", d["value"], d["ms_per_step"])
PY
