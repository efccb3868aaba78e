"""Gaussian linreg p = 3 just beyond the operand registers of mh_sweep_mfma (n = 10,000 .. 20,000), kernel_normal and kernel_adapt:
us per step and the kernel picked.  python tools/bench_ext.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi
C, steps = 1024, 3000
for n in (10000, 10300, 12000, 20000):
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, 3)); y = 1.0 + X @ np.linspace(1, -1, 3) + 4 * rng.standard_normal(n)
    init = np.array([0.0] * 4 + [y.std()])[None, :] + 0.05 * rng.standard_normal((C, 5)); init[:, -1] = np.abs(init[:, -1])
    gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
    big = E.DBL_MAX
    for kind, name in ((abi.KERNEL_NORMAL, "normal"), (abi.KERNEL_ADAPT, "adapt")):
        gk = E.KernelSpec(kind, 5, np.zeros(5), np.full(5, .02), np.full(5, -big), np.full(5, big), np.zeros(5, np.uint8), warmup=(500 if kind == abi.KERNEL_ADAPT else 0))
        best = 1e9
        for _ in range(3):
            st = E.ChainState(init, 5)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            E.sweep(gm, gk, st, steps, want_bits=False, check=False)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / steps)
        print("n=%6d %-6s %-14s %.3f us per step  frac %.3f" % (n, name, abi.last_kernel(), best, C / best * 1e6 * n * 9 / 78.6e12))
