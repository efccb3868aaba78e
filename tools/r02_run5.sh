set -o pipefail
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_multirank.py -x -q > gpurun_out/r02_gpu_tests_e.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r02_gpu_tests_e.log
