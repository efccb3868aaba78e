set -o pipefail
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "sharded or wide" > gpurun_out/r02_gpu_tests_k.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02_gpu_tests_k.log
timeout -k 10 120 python tools/bench_c4.py 400 > gpurun_out/r02_c4_e.log 2>&1; grep -v amdgpu.ids gpurun_out/r02_c4_e.log
FMCMC_AMD_LIB=/root/repo/fmcmc_amd/lib/libfmcmc_amd_stamp.so timeout -k 10 120 python tools/stamp_wide2.py 4 > gpurun_out/r02_stamp_w2.log 2>&1; grep -v amdgpu.ids gpurun_out/r02_stamp_w2.log
