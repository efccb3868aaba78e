"""Logistic model under kernel_adapt / kernel_ram on the observation-sharded sweep: the register owner in the hand-overs' shadow
(mh_sweep_logit2a, default) against the general kernel's sharded form (knob shadow=0).  C5's data (n = 100,000, k = 6) and n = 30,000."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi
big = E.DBL_MAX
nst = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for n, p in ((30000, 5), (100000, 5)):
    rng = np.random.default_rng(3 + n)
    X = rng.standard_normal((n, p)); beta = np.linspace(0.5, -0.5, p + 1)
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-(beta[0] + X @ beta[1:])))).astype(np.float64)
    gm = E.DeviceModel(abi.FAM_LOGISTIC, X, y, intercept=True, guard=False, prior_div=8.0)
    k = p + 1
    for kind, name in ((abi.KERNEL_ADAPT, "adapt"), (abi.KERNEL_RAM, "ram")):
        gk = E.KernelSpec(kind, k, np.zeros(k), np.ones(k), np.full(k, -big), np.full(k, big), np.zeros(k, np.uint8), warmup=50)
        for C in (512, 1024):
            init = beta[None, :] + 0.01 * rng.standard_normal((C, k))
            best = 1e9
            for _ in range(3):
                st = E.ChainState(init, gk.kf)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); r = E.sweep(gm, gk, st, nst, thin=10, seed=11, want_bits=False, check=False); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / nst)
            print("%-10s n=%-6d %-5s chains %4d: %7.2f us/step on %s" % (os.environ.get("FMCMC_AMD_DEBUG", "default"), n, name, C, best, abi.last_kernel()), flush=True)
