set -o pipefail
cd /root/repo
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests_g.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02_gpu_tests_g.log
python bench.py --config c2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r02_bench_c2_b.json 2> gpurun_out/r02_bench_c2_b.err; python -c "
import json; d=json.loads(open('gpurun_out/r02_bench_c2_b.json').read().strip().splitlines()[-1]); print('c2 value %.4e kernel_ms %.3f frac %.4f' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
python tools/bench_shapes.py > gpurun_out/r02_shapes_b.log 2>&1; grep -v amdgpu.ids gpurun_out/r02_shapes_b.log | tail -12
