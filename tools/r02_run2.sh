set -o pipefail
cd /root/repo
python tools/bench_c4.py 400 > gpurun_out/r02_c4_b.log 2>&1 && FMCMC_AMD_DEBUG=shard_mfma=0 python tools/bench_c4.py 400 > gpurun_out/r02_c4_b_valu.log 2>&1 && python tools/bench_c4.py 400 512 26 > gpurun_out/r02_c4_b_k26.log 2>&1 && python tools/bench_c4.py 400 512 64 > gpurun_out/r02_c4_b_k64.log 2>&1
cat gpurun_out/r02_c4_b.log gpurun_out/r02_c4_b_valu.log gpurun_out/r02_c4_b_k26.log gpurun_out/r02_c4_b_k64.log | grep -v amdgpu.ids
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests_c.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02_gpu_tests_c.log
