set -o pipefail
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "logistic" > gpurun_out/r02_gpu_tests_j.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02_gpu_tests_j.log
python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_c5_b.json 2> gpurun_out/r02_bench_c5_b.err; python -c "
import json; d=json.loads(open('gpurun_out/r02_bench_c5_b.json').read().strip().splitlines()[-1]); print('c5 value %.4e kernel_ms %.3f frac %.4f' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
