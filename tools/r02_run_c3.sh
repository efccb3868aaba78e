set -e
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "adapt or ram or spec or readme or c3 or continuation" > gpurun_out/c3_tests.log 2>&1 || { tail -30 gpurun_out/c3_tests.log; exit 1; }
tail -2 gpurun_out/c3_tests.log
for c in c3; do
timeout -k 10 300 python bench.py --config $c --steps 3 --warmup 1 > gpurun_out/c3_bench.json 2> gpurun_out/c3_bench.err
python - <<PY
import json; d=json.loads(open("gpurun_out/c3_bench.json").read().strip().splitlines()[-1]); print("$c", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
PY
done
