set -e
mkdir -p gpurun_out/r03
timeout -k 10 120 tools/exp_bin/probe_mfma4 > gpurun_out/r03/probe_mfma4.txt 2>&1
cat gpurun_out/r03/probe_mfma4.txt
for c in c2 c3 c4 c5; do
  timeout -k 10 300 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r03/base_$c.json 2> gpurun_out/r03/base_$c.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03/base_$c.json").read().strip().splitlines()[-1])
print("$c", d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"])
PY
done
timeout -k 10 200 python tools/stamp_mfma.py > gpurun_out/r03/stamp_mfma_base.txt 2>&1
cat gpurun_out/r03/stamp_mfma_base.txt
