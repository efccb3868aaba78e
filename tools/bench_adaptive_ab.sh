#!/bin/bash
# same-box A/B of the adaptive owner implementations (LDS vs register rows) for C3 / kernel_ram k=5
for r in 1 2; do
  echo "register rows:"; timeout 200 python tools/bench_configs.py 4000 2>&1 | grep -E "^C3|kernel_ram    linreg n=10k k=5"
  echo "LDS version (FMCMC_AMD_DEBUG=mode=16):"; FMCMC_AMD_DEBUG=mode=16 timeout 200 python tools/bench_configs.py 4000 2>&1 | grep -E "^C3|kernel_ram    linreg n=10k k=5"
done
