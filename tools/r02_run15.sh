set -o pipefail
cd /root/repo
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_perf_guard.py > gpurun_out/r02_gpu_tests_l.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02_gpu_tests_l.log
FMCMC_PERF_GUARD_RECORD=1 timeout -k 10 300 python -m pytest tests/test_gpu_perf_guard.py -x -q > gpurun_out/r02_perf_guard.log 2>&1; tail -2 gpurun_out/r02_perf_guard.log; cat gpurun_out/perf_guard.json | tr -d '\n '; echo
python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_c4_c.json 2> gpurun_out/r02_bench_c4_c.err; python -c "
import json; d=json.loads(open('gpurun_out/r02_bench_c4_c.json').read().strip().splitlines()[-1]); r=d['roofline']; print('c4 value %.4e ms/step %.1f kernel_ms %.2f checks %.2f frac %.4f kernel %s' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['gelman_checks_ms_per_step'], r['frac'], r['kernel']))"
