set -e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sharded or dataflow" > gpurun_out/w2_tests.log 2>&1 || { tail -30 gpurun_out/w2_tests.log; exit 1; }
tail -2 gpurun_out/w2_tests.log
timeout -k 10 300 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/w2_c4.json 2> gpurun_out/w2_c4.err
python - <<PY
import json; d=json.loads(open("gpurun_out/w2_c4.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])
PY
FMCMC_AMD_LIB=fmcmc_amd/lib/libfmcmc_amd_stamp.so timeout -k 10 150 python tools/stamp_wide2_events.py 2>&1 | tail -13
