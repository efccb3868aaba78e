#!/bin/bash
# C5 on the observation-sharded form (default) and on the chain-sharded loop (knob shard=0), same box
mkdir -p gpurun_out
python bench.py --config c5 --steps 4 --no-cpu-baseline > gpurun_out/c5_shard.json 2>/dev/null || exit 1
FMCMC_AMD_DEBUG=shard=0 python bench.py --config c5 --steps 3 --no-cpu-baseline --any-kernel > gpurun_out/c5_chain.json 2>/dev/null || exit 1
python - <<'PY'
import json
for f in ("gpurun_out/c5_shard.json", "gpurun_out/c5_chain.json"):
    d = json.load(open(f)); r = d["roofline"]
    print(f, "%.3e samples/s  %.2f us/step  kernel=%s  frac=%.3f" % (d["value"], 1e3 * r["kernel_ms"] / 4999, r["kernel"], r["frac"]))
PY
