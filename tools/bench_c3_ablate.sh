#!/bin/bash
# C3 timing ablations of the adaptive owner (results invalid on purpose): 2048 no Cholesky, 4096 no recursive mean / covariance
for m in 0 2048 4096 6144; do
  FMCMC_AMD_DEBUG=mode=$m python bench.py --config c3 --steps 20 --no-cpu-baseline --no-default-path --any-kernel 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('mode=$m  %.3f us/step  kernel=%s accept %.3f' % (1e3*r['kernel_ms']/9999, r['kernel'], d['config']['accept_rate']))"
done
