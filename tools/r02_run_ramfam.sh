set -e
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "families" > gpurun_out/ramfam_tests.log 2>&1 || { tail -30 gpurun_out/ramfam_tests.log; exit 1; }
tail -3 gpurun_out/ramfam_tests.log
