"""Scratch timing of the headline config (not the contract bench)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi
C, n, nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 10000, int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = np.random.default_rng(20260102)
X = rng.standard_normal((n, 3)); beta = np.array([3, 2, -1, .5]); y = beta[0] + X @ beta[1:] + 4 * rng.standard_normal(n)
init = np.array([0, 0, 0, 0, y.std()])[None, :] + 0.1 * rng.standard_normal((C, 5)); init[:, 4] = np.abs(init[:, 4])
gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
k = 5
gk = E.KernelSpec(abi.KERNEL_NORMAL, k, np.zeros(k), np.full(k, .02), np.full(k, -E.DBL_MAX), np.full(k, E.DBL_MAX), np.zeros(k, np.uint8))
for rep in range(3):
    st = E.ChainState(init, k)
    torch.cuda.synchronize(); t = time.time()
    r = E.sweep(gm, gk, st, nsteps, want_draws=False, want_logpost=False, want_bits=False)
    torch.cuda.synchronize(); dt = time.time() - t
    print("C=%d nsteps=%d: %.3f s  -> %.3e samples/s ; acc rate %.3f" % (C, nsteps, dt, C * (nsteps - 1) / dt, r.accept_count.double().mean().item() / (nsteps - 1)))
