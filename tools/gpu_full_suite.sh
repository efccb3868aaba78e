set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/full_gpu_tests.log 2>&1 || { tail -30 gpurun_out/full_gpu_tests.log; exit 1; }
tail -3 gpurun_out/full_gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
