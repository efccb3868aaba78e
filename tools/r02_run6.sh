set -o pipefail
cd /root/repo
python tools/bench_gelman.py > gpurun_out/r02_gelman_b.log 2>&1; grep -v amdgpu.ids gpurun_out/r02_gelman_b.log
timeout -k 10 900 python -m pytest tests/test_gpu_api.py -x -q -k "gelman or autostop" > gpurun_out/r02_gpu_tests_f.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r02_gpu_tests_f.log
