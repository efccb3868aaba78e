set -e
mkdir -p gpurun_out/r03
for r in 1 2; do
for L in libfmcmc_amd.so libfmcmc_amd_prio.so libfmcmc_amd_prio1.so; do
  echo "== $L"
  FMCMC_AMD_LIB=$PWD/fmcmc_amd/lib/$L timeout -k 10 200 python tools/bench_c4.py 1000 2>&1 | grep -v amdgpu.ids
  FMCMC_AMD_LIB=$PWD/fmcmc_amd/lib/$L timeout -k 10 200 python bench.py --config c3 --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', d['value'], d['roofline']['kernel_ms'])"
done; done 2>&1 | tee gpurun_out/r03/prio_ab.txt
