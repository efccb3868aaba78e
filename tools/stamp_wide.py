"""Diagnostic: where a step of the wide (observation-sharded) sweep goes, from s_memtime stamps of wave 0 of every workgroup.
Needs the stamped build:  python -c "from fmcmc_amd import build as b; b.build(extra_flags=['-DFMCMC_STAMP'], out='fmcmc_amd/lib/libfmcmc_amd_stamp.so')"
   FMCMC_AMD_LIB=fmcmc_amd/lib/libfmcmc_amd_stamp.so python tools/stamp_wide.py [kind=4] [K=50] [chains=512]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi
big = E.DBL_MAX
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
CH = int(sys.argv[3]) if len(sys.argv) > 3 else 512
n, nst = 10000, 400
rng = np.random.default_rng(20260104)
X4 = rng.standard_normal((n, K - 2)); b4 = rng.standard_normal(K - 1); y4 = b4[0] + X4 @ b4[1:] + 2 * rng.standard_normal(n)
init4 = np.concatenate([b4, [2.0]])[None, :] + 0.01 * rng.standard_normal((CH, K)); init4[:, -1] = np.abs(init4[:, -1])
z, o = np.zeros(K), np.ones(K)
gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X4, y4)
gk = E.KernelSpec(kind, K, z, o * (0.002 if kind == 1 else 1.0), -big * o, big * o, np.zeros(K, np.uint8))
import time
for _ in range(2):
    st = E.ChainState(init4, gk.kf)
    torch.cuda.synchronize(); t0 = time.time()
    r = E.sweep(gm, gk, st, nst, seed=1215, want_bits=False, check=False)
    torch.cuda.synchronize(); wall = (time.time() - t0) / nst * 1e6
print("kernel:", abi.last_kernel())
d = r.status_theta.cpu().numpy()[::2, :16] / (nst - 1)     # wave 0 = first chain of every workgroup; ticks per step
tps = np.median(d[:, :16].sum(axis=1)) / wall              # ticks per us, from the wall time of the (stamped) sweep
d = d / tps
print("wall %.1f us per step (stamped build), %.0f ticks per us" % (wall, tps))
names = ["rng tile", "proposal (A)", "sync", "publish", "barrier 1", "columns", "barrier 2", "gather+wave sum", "sync",
         "RAM adapt (B): rest", "accept/store (C): rest + tail", "B: logpost, exp, eta", "B: scan, cp", "B: coef", "B: rows", "C: logpost"]
print("us per step: median / min / max over workgroups")
tot = 0.0
for i, nm in enumerate(names):
    col = d[:, i]
    print("%-26s %7.2f %7.2f %7.2f" % (nm, np.median(col), col.min(), col.max()))
    tot += np.median(col)
print("%-26s %7.2f" % ("sum of medians", tot))
