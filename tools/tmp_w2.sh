set -e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sharded or dataflow" > gpurun_out/w2_tests.log 2>&1 || { tail -30 gpurun_out/w2_tests.log; exit 1; }
tail -2 gpurun_out/w2_tests.log
for t in 1 0; do
FMCMC_AMD_DEBUG=tiles=$t timeout -k 10 300 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/w2_c4.json 2> gpurun_out/w2_c4.err
python - <<PY
import json; d=json.loads(open("gpurun_out/w2_c4.json").read().strip().splitlines()[-1]); print("tiles=$t", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])
PY
done
