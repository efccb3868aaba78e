"""Diagnostic: per-wave cycle shares of mh_sweep_mfma (FMCMC_AMD_DEBUG=mode=8,mfma=1)."""
import os, sys
os.environ["FMCMC_AMD_DEBUG"] = "mode=8,mfma=1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi
C, n, nsteps = 1024, 10000, 3000
rng = np.random.default_rng(20260102)
X = rng.standard_normal((n, 3)); y = 3 + X @ np.array([2, -1, .5]) + 4 * rng.standard_normal(n)
init = np.array([0, 0, 0, 0, y.std()])[None, :] + 0.1 * rng.standard_normal((C, 5)); init[:, 4] = np.abs(init[:, 4])
gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
gk = E.KernelSpec(1, 5, np.zeros(5), np.full(5, .02), np.full(5, -E.DBL_MAX), np.full(5, E.DBL_MAX), np.zeros(5, np.uint8))
st = E.ChainState(init, 5)
r = E.sweep(gm, gk, st, nsteps, want_draws=False, want_logpost=True, check=False)
torch.cuda.synchronize()
flat = r.logpost.reshape(-1).cpu().numpy()
nb = C // 4
rows = np.array([[flat[flat.size - 8 * (b * 8 + w + 1): flat.size - 8 * (b * 8 + w + 1) + 8] for w in range(8)] for b in range(nb)])
per = rows[:, :, [0, 1, 2, 3, 5, 6, 7]] / rows[:, :, 4:5]
med = np.median(per, axis=0)
print("ticks per MH step, median over workgroups: eval(MFMA) | barrier wait | owner phase | barrier wait | total || owner phase split: fold | closed form | decide (rest = propose)")
for w in range(8):
    print("wave %d: %7.0f %7.0f %7.0f %7.0f | %7.0f || %6.0f %6.0f %6.0f" % (w, *med[w][:4], med[w][:4].sum(), *med[w][4:]))
