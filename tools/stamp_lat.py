"""Diagnostic: per-wave cycle shares of mh_sweep_spec in the latency form (FMCMC_AMD_DEBUG=mode=8). Not a benchmark.
   python tools/stamp_lat.py [chains=256]"""
import os, sys
os.environ["FMCMC_AMD_DEBUG"] = "mode=8"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n, nsteps = 10000, 3000
rng = np.random.default_rng(20260102)
X = rng.standard_normal((n, 3)); y = 3 + X @ np.array([2, -1, .5]) + 4 * rng.standard_normal(n)
init = np.array([0, 0, 0, 0, y.std()])[None, :] + 0.1 * rng.standard_normal((C, 5)); init[:, 4] = np.abs(init[:, 4])
gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
gk = E.KernelSpec(1, 5, np.zeros(5), np.full(5, .02), np.full(5, -E.DBL_MAX), np.full(5, E.DBL_MAX), np.zeros(5, np.uint8))
st = E.ChainState(init, 5)
r = E.sweep(gm, gk, st, nsteps, want_draws=True, check=False)
torch.cuda.synchronize()
kn = abi.last_kernel().split(":")[-1]
cw = int(kn[-1]) if kn.startswith("spec-lat") else 4
nb = (C + cw - 1) // cw
d = r.draws.reshape(-1)[: nb * 12 * 4].cpu().numpy().reshape(nb, 12, 4)
per = d[:, :, :3] / np.maximum(d[:, :, 3:4], 1)
print("kernel %s: ticks per MH step (s_memtime), median over %d workgroups" % (kn, nb))
print("waves 0-7 compute = (flag wait, eval, -), waves 8-11 owners = (flag wait, phase to publish, stores)")
med = np.median(per, axis=0)
for w in range(12):
    print("wave %2d: %8.0f %8.0f %8.0f | %8.0f" % (w, med[w, 0], med[w, 1], med[w, 2], med[w].sum()))
