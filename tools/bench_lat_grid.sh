#!/bin/bash
# the latency form against the four-chains-per-workgroup kernels over shapes: n x p x chain counts (tools/bench_chains.py)
for kind in normal adapt; do
  for np in "600 1" "1000 3" "2500 3" "5000 3" "10000 3" "5000 5" "4000 7"; do
    set -- $np
    BENCH_N=$1 BENCH_P=$2 python tools/bench_chains.py $kind 3000 64 256 512 768 2>/dev/null | grep -v "^{"
  done
done
