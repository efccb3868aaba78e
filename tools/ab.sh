#!/bin/bash
# same-box A/B of library builds: tools/ab.sh <steps> libA libB ... (interleaved rounds)
N=${1:-10000}; shift
for round in 1 2 3; do
  for L in "$@"; do
    echo -n "$(basename $L): "
    FMCMC_AMD_LIB=$PWD/$L timeout 120 python tools/quick_bench.py 1024 $N 2>&1 | tail -1
  done
done
