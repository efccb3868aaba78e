set -e
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "step_windows or longer_than_4_gib or full_size_headline" > gpurun_out/r03/window_tests.log 2>&1 || { tail -40 gpurun_out/r03/window_tests.log; exit 1; }
tail -3 gpurun_out/r03/window_tests.log
python tools/bench_long.py 2>&1 | grep -v amdgpu.ids
