"""Diagnostic: per-wave cycle shares of mh_sweep_spec for kernel_adapt (KIND=3) or kernel_ram (KIND=4)."""
import os, sys
os.environ["FMCMC_AMD_DEBUG"] = "mode=8"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi
C, n, nsteps = 1024, 10000, 3000
rng = np.random.default_rng(20260102)
X = rng.standard_normal((n, 3)); y = 3 + X @ np.array([2, -1, .5]) + 4 * rng.standard_normal(n)
init = np.array([0, 0, 0, 0, y.std()])[None, :] + 0.1 * rng.standard_normal((C, 5)); init[:, 4] = np.abs(init[:, 4])
gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
KIND = int(os.environ.get("KIND", "3"))
gk = E.KernelSpec(KIND, 5, np.zeros(5), np.full(5, .02), np.full(5, -E.DBL_MAX), np.full(5, E.DBL_MAX), np.zeros(5, np.uint8), warmup=500 if KIND == 3 else 0)
st = E.ChainState(init, 5)
r = E.sweep(gm, gk, st, nsteps, want_draws=True, check=False)
torch.cuda.synchronize()
NWV = 8 if "spec=0" in os.environ.get("FMCMC_AMD_DEBUG", "") else 12
d = r.draws.reshape(-1)[: (C // 4) * NWV * 4].cpu().numpy().reshape(C // 4, NWV, 4)
per = d[:, :, :3] / d[:, :, 3:4]
print("cycles per MH step (s_memtime ticks), median over workgroups")
print("spec kernel: waves 0-7 compute = (flag wait, eval, -), waves 8-11 owners = (flag wait, phase to publish, stores)")
med = np.median(per, axis=0)
for w in range(NWV):
    print("wave %2d: %8.0f %8.0f %8.0f | %8.0f" % (w, med[w, 0], med[w, 1], med[w, 2], med[w].sum()))
