set -e
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_shim.py tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r03/shim_api_tests.log 2>&1 || { tail -40 gpurun_out/r03/shim_api_tests.log; exit 1; }
tail -3 gpurun_out/r03/shim_api_tests.log
