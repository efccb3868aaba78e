"""Time per MH step against the number of chains on ONE GPU, at C2 / C3's shape (n = 10,000, k = 5): what a GPU of a
strong-scaled call sees (1024 chains over G GPUs = 1024 / G each).  For every chain count the dispatcher's own choice and the
four-chains-per-workgroup kernels (knob lat=0) side by side.

  python tools/bench_chains.py [normal|adapt|ram] [steps] [chain counts ...]   -> one line per count + a JSON summary
  BENCH_N / BENCH_P in the environment: another shape (n observations, p covariates) instead of C2's"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fmcmc_amd import engine as E, _abi as abi  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "normal"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
counts = [int(a) for a in sys.argv[3:]] or [4, 64, 128, 256, 257, 384, 512, 513, 640, 768, 769, 1024]
cfg = bench.Config({"normal": "c2", "adapt": "c3", "ram": "c3"}[kind])
if "BENCH_N" in os.environ:
    cfg.n, cfg.p = int(os.environ["BENCH_N"]), int(os.environ.get("BENCH_P", "3"))
    cfg.k = cfg.p + 2
    rng = np.random.default_rng(5)
    X = rng.standard_normal((cfg.n, cfg.p))
    y = 1.0 + X @ np.linspace(1.0, -1.0, cfg.p) + 4.0 * rng.standard_normal(cfg.n)
    init = np.concatenate([np.zeros(cfg.p + 1), [float(np.std(y))]])[None, :] + 0.05 * rng.standard_normal((max(counts), cfg.k))
    init[:, -1] = np.abs(init[:, -1])
else:
    X, y, init = cfg.workload(max(counts), 0)
K = cfg.k
gm, gk = bench.device_objects(cfg, E, abi, X, y, "cuda:0")
if kind == "ram":
    big = E.DBL_MAX
    gk = E.KernelSpec(abi.KERNEL_RAM, K, np.zeros(K), np.ones(K), np.full(K, -big), np.full(K, big), np.zeros(K, np.uint8))


def us_per_step(C):
    init_d = torch.as_tensor(init[:C]).cuda()
    best = None
    for rep in range(3):
        st = E.ChainState(init_d, K)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = E.sweep(gm, gk, st, steps, seed=bench.CHAIN_SEED, want_logpost=True, want_draws=True, want_bits=False, check=False)
        e1.record()
        torch.cuda.synchronize()
        assert int(r.status.abs().sum().item()) == 0
        t = e0.elapsed_time(e1) * 1e3 / (steps - 1)
        best = t if best is None else min(best, t)
    return best, abi.last_kernel()


rows = []
for C in counts:
    os.environ.pop("FMCMC_AMD_DEBUG", None)
    a, ka = us_per_step(C)
    os.environ["FMCMC_AMD_DEBUG"] = "lat=0"
    b, kb = us_per_step(C)
    os.environ.pop("FMCMC_AMD_DEBUG", None)
    rows.append({"chains": C, "us_per_step": round(a, 3), "kernel": ka, "us_per_step_lat0": round(b, 3), "kernel_lat0": kb,
                 "samples_per_s": round(C / a * 1e6)})
    print("%-6s n=%d p=%d chains %5d: %7.3f us/step on %-10s (four per workgroup: %7.3f on %s)  %.3e samples/s" % (kind, cfg.n, cfg.p, C, a, ka, b, kb, C / a * 1e6), flush=True)
print(json.dumps({"kind": kind, "steps": steps, "rows": rows}))
