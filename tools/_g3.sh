set -e
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "logistic or logit or uniform or mirror or iid or dispatcher or randomised" > gpurun_out/r03/logit_tests.log 2>&1 || { tail -30 gpurun_out/r03/logit_tests.log; exit 1; }
tail -5 gpurun_out/r03/logit_tests.log
