"""Throughput of Gaussian linreg shapes around the headline one: MFMA evaluation (default) vs the paths the same shapes
took before the MFMA kernel was made general in n and p (FMCMC_AMD_DEBUG=mfma=0)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi
C, nsteps = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for n, p in [(10000, 3), (8000, 3), (5000, 3), (2000, 3), (600, 3), (10000, 1), (10000, 2), (5000, 5), (3000, 7), (1000, 1)]:
    rng = np.random.default_rng(n + p)
    X = rng.standard_normal((n, p)); y = 1.0 + X @ np.linspace(1, -1, p) + 4 * rng.standard_normal(n)
    k = p + 2
    init = np.array([0.0] * (p + 1) + [y.std()])[None, :] + 0.05 * rng.standard_normal((C, k)); init[:, -1] = np.abs(init[:, -1])
    gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
    gk = E.KernelSpec(abi.KERNEL_NORMAL, k, np.zeros(k), np.full(k, .02), np.full(k, -E.DBL_MAX), np.full(k, E.DBL_MAX), np.zeros(k, np.uint8))
    res = {}
    for mode in ("1", "0"):
        os.environ["FMCMC_AMD_DEBUG"] = "mfma=" + mode
        best = 0.0
        for rep in range(3):
            st = E.ChainState(init, k)
            torch.cuda.synchronize(); t = time.time()
            r = E.sweep(gm, gk, st, nsteps, want_draws=False, want_logpost=False, want_bits=False)
            torch.cuda.synchronize(); best = max(best, C * (nsteps - 1) / (time.time() - t))
        res[mode] = best
    print("n=%6d p=%d: MFMA %.3e samples/s | before %.3e | x%.2f" % (n, p, res["1"], res["0"], res["1"] / res["0"]))
