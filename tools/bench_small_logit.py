"""The workflow vignette's own size (mcmc::logit: 100 observations, 4 covariates + intercept; vignettes/workflow-with-fmcmc.Rmd:22-60) and
a few more small logistic shapes: time per MH step against the number of chains, kernel_normal and kernel_adapt."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi
big = E.DBL_MAX
nst = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
for n, p in ((100, 4), (1000, 4), (5000, 4)):
    rng = np.random.default_rng(7 + n)
    X = rng.standard_normal((n, p)); beta = np.array([0.5, 1.0, -1.0, 0.5, 0.25])
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-(beta[0] + X @ beta[1:])))).astype(np.float64)
    gm = E.DeviceModel(abi.FAM_LOGISTIC, X, y, intercept=True, guard=False, prior_div=8.0)
    k = p + 1
    for kind, name in ((abi.KERNEL_NORMAL, "normal"), (abi.KERNEL_ADAPT, "adapt")):
        gk = (E.KernelSpec(kind, k, np.zeros(k), np.full(k, 0.2), np.full(k, -big), np.full(k, big), np.zeros(k, np.uint8)) if kind == abi.KERNEL_NORMAL
              else E.KernelSpec(kind, k, np.zeros(k), np.ones(k), np.full(k, -big), np.full(k, big), np.zeros(k, np.uint8), warmup=500))
        for C in (1, 2, 4, 64, 256, 1024):
            init = beta[None, :] + 0.05 * rng.standard_normal((C, k))
            best = 1e9
            for _ in range(3):
                st = E.ChainState(init, gk.kf)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = E.sweep(gm, gk, st, nst, seed=11, want_bits=False, check=False)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / nst)
            print("logistic n=%-5d p=%d %-6s chains %5d: %7.3f us/step on %-20s %.3e samples/s" % (n, p, name, C, best, abi.last_kernel(), C * 1e6 / best), flush=True)
