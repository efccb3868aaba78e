set -o pipefail
cd /root/repo
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_perf_guard.py > gpurun_out/r02_gpu_tests_m.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02_gpu_tests_m.log
FMCMC_PERF_GUARD_RECORD=1 timeout -k 10 300 python -m pytest tests/test_gpu_perf_guard.py -x -q > gpurun_out/r02_perf_guard.log 2>&1; tail -2 gpurun_out/r02_perf_guard.log; cat gpurun_out/perf_guard.json | tr -d '\n '; echo
