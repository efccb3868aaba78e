"""Diagnostic: WHEN the hand-overs of one step of the dataflow wide sweep (mh_sweep_wide2) happen, on the chip-wide clock
(s_memrealtime, 100 MHz): publish -> proposals seen -> partials arrived -> partials seen, per chain group, spread over the
256 workgroups.  Needs the stamped build (tools/stamp_wide.py says how); FMCMC_AMD_LIB=...stamp.so python tools/stamp_wide2_events.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi
big = E.DBL_MAX
K, CH, n, nst = 50, 512, 10000, 400
rng = np.random.default_rng(20260104)
X4 = rng.standard_normal((n, K - 2)); b4 = rng.standard_normal(K - 1); y4 = b4[0] + X4 @ b4[1:] + 2 * rng.standard_normal(n)
init4 = np.concatenate([b4, [2.0]])[None, :] + 0.01 * rng.standard_normal((CH, K)); init4[:, -1] = np.abs(init4[:, -1])
z, o = np.zeros(K), np.ones(K)
gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X4, y4)
gk = E.KernelSpec(4, K, z, o, -big * o, big * o, np.zeros(K, np.uint8))
st = E.ChainState(init4, gk.kf)
r = E.sweep(gm, gk, st, nst, seed=1215, want_bits=False, check=False)
torch.cuda.synchronize()
print("kernel:", abi.last_kernel())
d = r.status_theta.cpu().numpy()[::2] * 0.01          # us
# a workgroup whose stamped wave never passed one of the stamps (a slot left at 0: the sweep ended first, or the wave is not
# the stamped one in that workgroup) stays out of the statistics -- an unset stamp used to enter them as "0 us", i.e. as
# minus the whole run time relative to the median publish
cols = [24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 36, 37]
ok = (d[:, cols] > 0).all(axis=1)
print("workgroups with every stamp set: %d of %d" % (int(ok.sum()), d.shape[0]))
d = d[ok]
if d.shape[0] == 0:
    raise SystemExit("no workgroup has all stamps: is FMCMC_AMD_LIB the -DFMCMC_STAMP build?")
for g in (0, 1):
    pub300, x1seen, x2arr, x2arr5, x2seen, pub301 = d[:, 32 + g], d[:, 26 + 2 * g], d[:, 27 + 2 * g], d[:, 36 + g], d[:, 24 + 6 * g], d[:, 25 + 6 * g]
    t0 = np.median(pub300)
    f = lambda a: "%7.2f %7.2f %7.2f" % (a.min() - t0, np.median(a) - t0, a.max() - t0)
    print("group %d (us after the median publish of version 300; min / median / max over workgroups)" % g)
    print("  owners published v300        ", f(pub300))
    print("  evaluators saw the proposals ", f(x1seen), "  -> hand-over after the LAST publish: %.2f .. %.2f" % (x1seen.min() - pub300.max(), x1seen.max() - pub300.max()))
    print("  wave 2 past its arrival      ", f(x2arr), "  (wave 5: %s)" % f(x2arr5))
    last = np.maximum(x2arr, x2arr5).max()
    print("  owners saw the partials      ", f(x2seen), "  -> hand-over after the LAST arrival: %.2f .. %.2f" % (x2seen.min() - last, x2seen.max() - last))
    print("  owners published v301        ", f(pub301), "  cycle %.2f" % (np.median(pub301) - t0))
