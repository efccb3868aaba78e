set -o pipefail
cd /root/repo
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests_d.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02_gpu_tests_d.log
python tools/bench_c4.py 400 > gpurun_out/r02_c4_d.log 2>&1; grep -v amdgpu.ids gpurun_out/r02_c4_d.log
python tools/bench_configs.py 10000 > gpurun_out/r02_cfg_d.log 2>&1; grep -v amdgpu.ids gpurun_out/r02_cfg_d.log
FMCMC_AMD_LIB=/root/repo/fmcmc_amd/lib/libfmcmc_amd_stamp.so python tools/stamp_wide.py 4 50 > gpurun_out/r02_stamp_ram.log 2>&1; grep -v amdgpu.ids gpurun_out/r02_stamp_ram.log
