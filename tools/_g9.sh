set -e
python tools/bench_long.py 2>&1 | grep -v amdgpu.ids
FMCMC_AMD_DEBUG=window=8192 python tools/bench_long.py 2>&1 | grep -v amdgpu.ids
FMCMC_AMD_DEBUG=window=320 python tools/bench_long.py 2>&1 | grep -v amdgpu.ids
