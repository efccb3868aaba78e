"""Diagnostic: where a step of the dataflow wide sweep (mh_sweep_wide2) goes: s_memtime shares of owner wave 0 and evaluator
wave 2 of every workgroup.  Needs the stamped build (tools/stamp_wide.py says how); FMCMC_AMD_LIB=...stamp.so python tools/stamp_wide2.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi
big = E.DBL_MAX
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K, CH, n, nst = 50, 512, 10000, 400
rng = np.random.default_rng(20260104)
X4 = rng.standard_normal((n, K - 2)); b4 = rng.standard_normal(K - 1); y4 = b4[0] + X4 @ b4[1:] + 2 * rng.standard_normal(n)
init4 = np.concatenate([b4, [2.0]])[None, :] + 0.01 * rng.standard_normal((CH, K)); init4[:, -1] = np.abs(init4[:, -1])
z, o = np.zeros(K), np.ones(K)
gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X4, y4)
gk = E.KernelSpec(kind, K, z, o * (0.002 if kind == 1 else 1.0), -big * o, big * o, np.zeros(K, np.uint8), until=float(os.environ.get("STAMP_UNTIL", "inf")))
for _ in range(2):
    st = E.ChainState(init4, gk.kf)
    torch.cuda.synchronize(); t0 = time.time()
    r = E.sweep(gm, gk, st, nst, seed=1215, want_bits=False, check=False)
    torch.cuda.synchronize(); wall = (time.time() - t0) / nst * 1e6
print("kernel:", abi.last_kernel(), " wall %.1f us per step (stamped build)" % wall)
d = r.status_theta.cpu().numpy()[::2] / (nst - 1)
ow, ev = d[:, :8], d[:, 16:20]
tps = np.median(ow[:, :7].sum(axis=1)) / wall
print("owner wave 0 (us per step, median / max over workgroups); %.0f ticks per us" % tps)
for i, nm in enumerate(["wait for partials", "gather + log-posterior", "adapt (ram)", "accept + store", "propose", "publish (drain + arrive)", "draw variates"]):
    print("  %-26s %7.2f %7.2f" % (nm, np.median(ow[:, i]) / tps, ow[:, i].max() / tps))
print("evaluator wave 2 (us per step = both groups)")
for i, nm in enumerate(["wait for proposals", "slice product (MFMA)", "drain stores", "arrive"]):
    print("  %-26s %7.2f %7.2f" % (nm, np.median(ev[:, i]) / tps, ev[:, i].max() / tps))
if os.environ.get("STAMP_ALL", "0") == "1":      # the build with -DFMCMC_STAMP_WAVE=-1: every evaluator wave (2 .. 7; SIMD = wave % 4)
    print("all evaluator waves (us per step: wait for proposals | slice product | drain | arrive)")
    for w in range(2, 8):
        e = d[:, 16 + 4 * (w - 2):20 + 4 * (w - 2)]
        print("  wave %d (SIMD %d): %7.2f %7.2f %7.2f %7.2f" % ((w, w % 4) + tuple(np.median(e[:, i]) / tps for i in range(4))))
