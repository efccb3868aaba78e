set -e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sharded or dataflow or randomised" > gpurun_out/w2_tests.log 2>&1 || { tail -30 gpurun_out/w2_tests.log; exit 1; }
tail -2 gpurun_out/w2_tests.log
for g in 2 4; do
FMCMC_AMD_DEBUG=groups=$g timeout -k 10 300 python bench.py --config c4 --steps 3 --warmup 1 > gpurun_out/w2_c4.json 2> gpurun_out/w2_c4.err
python - <<PY
import json; d=json.loads(open("gpurun_out/w2_c4.json").read().strip().splitlines()[-1]); print("groups $g", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])
PY
done
FMCMC_AMD_LIB=fmcmc_amd/lib/libfmcmc_amd_stamp.so timeout -k 10 200 python tools/stamp_wide2.py > gpurun_out/stamp_w2.log 2>&1; tail -16 gpurun_out/stamp_w2.log
