set -o pipefail
cd /root/repo
export FMCMC_AMD_LIB=/root/repo/fmcmc_amd/lib/libfmcmc_amd_stamp.so
python tools/stamp_wide.py 4 50 > gpurun_out/r02_stamp_ram.log 2>&1 && python tools/stamp_wide.py 1 50 > gpurun_out/r02_stamp_normal.log 2>&1
cat gpurun_out/r02_stamp_ram.log gpurun_out/r02_stamp_normal.log | grep -v amdgpu.ids
