import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from fmcmc_amd import engine as E, _abi as abi
C, nsteps = 1024, 5000
for n, p in [(1000, 1), (600, 1), (1024, 1), (900, 1)]:
    rng = np.random.default_rng(n + p)
    X = rng.standard_normal((n, p)); y = 1.0 + X @ np.linspace(1, -1, p) + 4 * rng.standard_normal(n)
    k = p + 2
    init = np.array([0.0] * (p + 1) + [y.std()])[None, :] + 0.05 * rng.standard_normal((C, k)); init[:, -1] = np.abs(init[:, -1])
    gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
    for kind in (1, 2):
        lb = np.full(k, -E.DBL_MAX if kind == 1 else -10.0); ub = np.full(k, E.DBL_MAX if kind == 1 else 10.0)
        gk = E.KernelSpec(kind, k, np.zeros(k), np.full(k, .05), lb, ub, np.zeros(k, np.uint8))
        res = {}
        for mode in ("0", "1"):
            os.environ["FMCMC_AMD_DEBUG"] = "spec=" + ("0" if mode == "1" else "1")
            best = 0.0
            for rep in range(3):
                st = E.ChainState(init, k)
                torch.cuda.synchronize(); t = time.time()
                r = E.sweep(gm, gk, st, nsteps, want_draws=True, want_logpost=True, want_bits=False)
                torch.cuda.synchronize(); best = max(best, C * (nsteps - 1) / (time.time() - t))
            res[mode] = best
        print("n=%5d p=%d kind=%d: spec %.3e | mfma %.3e | x%.2f" % (n, p, kind, res["0"], res["1"], res["1"] / res["0"]))
