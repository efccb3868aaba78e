set -e
mkdir -p gpurun_out/r03
FMCMC_AMD_LIB=$PWD/fmcmc_amd/lib/libfmcmc_amd_ab.so timeout -k 10 200 python tools/stamp_mfma.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/stamp_mfma_r03.txt
FMCMC_AMD_LIB=$PWD/fmcmc_amd/lib/libfmcmc_amd_ab.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mfma_equals_valu or replicated" 2>&1 | tail -3
