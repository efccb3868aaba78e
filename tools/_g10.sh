set -e
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "bench_multi" > gpurun_out/r03/multirank_bench.log 2>&1 || { tail -40 gpurun_out/r03/multirank_bench.log; exit 1; }
tail -3 gpurun_out/r03/multirank_bench.log
