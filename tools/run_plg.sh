#!/bin/bash
# runs the logistic design probes built into tools/exp_bin (tools/probe_logit_grid.hip, tools/probe_logit_shard.hip)
mkdir -p gpurun_out
for b in tools/exp_bin/plg_* tools/exp_bin/pls_*; do [ -x "$b" ] && { timeout -k 10 120 $b 100000 40 || exit 1; }; done > gpurun_out/plg.txt 2>&1
cat gpurun_out/plg.txt
