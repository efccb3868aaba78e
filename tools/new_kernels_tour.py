"""One call per kernel path round 4 added, for a rocprofv3 --kernel-trace --stats pass (profiles/r04_new_kernels_*):
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d <out> -- python3 tools/new_kernels_tour.py
Prints, per call, the engine kernel and the time per step by HIP events (2000-step sweeps, 1024 or 4 chains)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi  # noqa: E402

rng = np.random.default_rng(4)
big = E.DBL_MAX


def call(label, fam, n, p, chains, kind, steps=2000, **kw):
    X = rng.standard_normal((n, p)) if p else None
    if fam == abi.FAM_LOGISTIC:
        beta = np.concatenate([[-1.0], np.linspace(.5, -.5, p)])
        y = (rng.uniform(size=n) < 1 / (1 + np.exp(-(beta[0] + X @ beta[1:])))).astype(np.float64)
        k = p + 1
        init = beta[None, :] + 0.01 * rng.standard_normal((chains, k))
        gm = E.DeviceModel(fam, X, y, intercept=True, guard=False, prior_div=8.0)
        scale = np.full(k, .01)
    else:
        y = 1.0 + (X @ np.linspace(1, -1, p) if p else 0.0) + 4 * rng.standard_normal(n)
        k = p + 2
        init = np.array([0.0] * (p + 1) + [y.std()])[None, :] + 0.05 * rng.standard_normal((chains, k)); init[:, -1] = np.abs(init[:, -1])
        gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG if p else abi.FAM_IID_NORMAL, X, y)
        scale = np.full(k, .02 / max(1.0, (p / 3.0) ** 0.5))
    lb = kw.pop("lb", np.full(k, -big)); ub = kw.pop("ub", np.full(k, big))
    gk = E.KernelSpec(kind, k, kw.pop("mu", np.zeros(k)), kw.pop("scale", scale), lb, ub, np.zeros(k, np.uint8), **kw)

    def go():
        st = E.ChainState(init, k)
        E.sweep(gm, gk, st, steps, thin=10, want_bits=False, check=False)
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(); e1.record(); torch.cuda.synchronize()
    print("%-62s %-28s %8.2f us / step" % (label, abi.last_kernel(), e0.elapsed_time(e1) / (steps - 1) * 1e3), flush=True)


L, G = abi.FAM_LOGISTIC, abi.FAM_GAUSSIAN_LINREG
lb5 = np.array([-big] * 4 + [0.001])
call("linreg n=20000 p=3, 1024 chains, kernel_normal", G, 20000, 3, 1024, abi.KERNEL_NORMAL)
call("linreg n=10000 p=12, 1024 chains, kernel_normal", G, 10000, 12, 1024, abi.KERNEL_NORMAL)
call("linreg n=20000 p=3, 1024 chains, kernel_adapt", G, 20000, 3, 1024, abi.KERNEL_ADAPT, warmup=500)
call("linreg n=10000 p=3, 1024 chains, kernel_ram(lb)", G, 10000, 3, 1024, abi.KERNEL_RAM, lb=lb5)
call("linreg n=10000 p=12, 1024 chains, kernel_adapt", G, 10000, 12, 1024, abi.KERNEL_ADAPT, warmup=500)
call("linreg n=10000 p=3, 1024 chains, kernel_nmirror", G, 10000, 3, 1024, abi.KERNEL_NMIRROR, mu=np.array([0, 0, 0, 0, 4.0]), scale=np.full(5, 0.1), warmup=500, nadapt=5)
call("linreg n=20000 p=48, 512 chains, kernel_normal", G, 20000, 48, 512, abi.KERNEL_NORMAL, steps=600)
call("linreg n=100000 p=3, 4 chains, kernel_normal", G, 100000, 3, 4, abi.KERNEL_NORMAL)
call("linreg n=1000000 p=3, 4 chains, kernel_adapt", G, 1000000, 3, 4, abi.KERNEL_ADAPT, warmup=500, steps=600)
call("linreg n=100000 p=48, 4 chains, kernel_ram", G, 100000, 48, 4, abi.KERNEL_RAM, steps=600)
call("iid Normal n=10000, 1024 chains, kernel_normal", G, 10000, 0, 1024, abi.KERNEL_NORMAL)
call("logistic n=30000 p=5, 1024 chains, kernel_adapt", L, 30000, 5, 1024, abi.KERNEL_ADAPT, warmup=500, steps=600)
call("logistic n=100000 p=12, 1024 chains, kernel_normal_reflective", L, 100000, 12, 1024, abi.KERNEL_NORMAL_REFLECTIVE, lb=np.full(13, -5.0), ub=np.full(13, 5.0), steps=400)
call("logistic n=100000 p=5, 4096 chains, kernel_normal_reflective", L, 100000, 5, 4096, abi.KERNEL_NORMAL_REFLECTIVE, lb=np.full(6, -5.0), ub=np.full(6, 5.0), steps=200)
call("logistic n=100000 p=5, 4 chains, kernel_normal_reflective", L, 100000, 5, 4, abi.KERNEL_NORMAL_REFLECTIVE, lb=np.full(6, -5.0), ub=np.full(6, 5.0))
