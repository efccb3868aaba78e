"""Streamed kernel, chains per workgroup (FMCMC_AMD_DEBUG=cw=N): L2 traffic per evaluation vs CUs kept busy."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi
big = E.DBL_MAX
n, C, nsteps = 10000, 512, 300
for k, kind in ((30, 4), (50, 1), (50, 4), (30, 1)):
    rng = np.random.default_rng(k)
    X = rng.standard_normal((n, k - 2)); b = rng.standard_normal(k - 1); y = b[0] + X @ b[1:] + 2 * rng.standard_normal(n)
    init = np.concatenate([b, [2.0]])[None, :] + 0.01 * rng.standard_normal((C, k)); init[:, -1] = np.abs(init[:, -1])
    gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
    z, o = np.zeros(k), np.ones(k)
    gk = E.KernelSpec(kind, k, z, o * 0.002, -big * o, big * o, np.zeros(k, np.uint8))
    for cw in ("1", "2", "4", "8"):
        os.environ["FMCMC_AMD_DEBUG"] = "cw=" + cw
        best = 1e9
        try:
            for _ in range(2):
                st = E.ChainState(init, k)
                torch.cuda.synchronize(); t = time.time()
                E.sweep(gm, gk, st, nsteps, seed=1, want_bits=False, want_draws=False, check=False)
                torch.cuda.synchronize(); best = min(best, time.time() - t)
            print("k=%d kind=%d CW=%s: %.1f us/step  %.3e samples/s" % (k, kind, cw, best / nsteps * 1e6, C * (nsteps - 1) / best))
        except Exception as ex:
            print("k=%d kind=%d CW=%s: %s" % (k, kind, cw, str(ex)[:60]))
