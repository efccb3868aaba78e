#!/bin/bash
# C5: the younger wave's turn at the issue priority (logit_shard, knob turn=<thousandths of its passes>), us per step
mkdir -p gpurun_out
for t in ${TURNS:-0 300 450 550 640 720 800 1000}; do
  FMCMC_AMD_DEBUG=turn=$t python bench.py --config c5 --steps ${STEPS:-3} --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('turn=$t  %.2f us/step  frac %.4f  kernel=%s' % (1e3*r['kernel_ms']/4999, r['frac'], r['kernel']))"
done
