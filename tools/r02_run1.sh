set -o pipefail
cd /root/repo

timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "sharded or wide" > gpurun_out/r02_gpu_tests_b.log 2>&1; echo "pytest sharded rc=$?"; tail -5 gpurun_out/r02_gpu_tests_b.log
