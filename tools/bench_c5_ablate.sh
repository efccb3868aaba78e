#!/bin/bash
# C5 timing ablations (results invalid on purpose): mode=32 skips the two grid barriers of a step, mode=128 the gather
mkdir -p gpurun_out
for m in 0 32 160; do
  FMCMC_AMD_DEBUG=mode=$m python bench.py --config c5 --steps 3 --no-cpu-baseline --any-kernel 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('mode=$m  %.2f us/step  kernel=%s' % (1e3*r['kernel_ms']/4999, r['kernel']))"
done
