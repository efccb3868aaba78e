"""Diagnostic: where a step of config C5 (logistic n = 100,000, 1024 chains, observation-sharded) goes, from s_memtime stamps of wave 0
of every workgroup.  Needs the stamped build:
   python -c "from fmcmc_amd import build as b; b.build(extra_flags=['-DFMCMC_STAMP'], out='fmcmc_amd/lib/libfmcmc_amd_stamp.so')"
   FMCMC_AMD_LIB=fmcmc_amd/lib/libfmcmc_amd_stamp.so python tools/stamp_c5.py
(extra_flags=['-DFMCMC_STAMP', '-DFMCMC_STAMP_WAVE=-1'] and STAMP_ALL=1: the shares of all eight waves)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fmcmc_amd import engine as E, _abi as abi
cfg = bench.Config("c5")
X, y, init = cfg.workload(cfg.chains, 0)
gm, gk = bench.device_objects(cfg, E, abi, X, y, "cuda:0")
nst = 400
for _ in range(2):
    st = E.ChainState(torch.as_tensor(init).cuda(), cfg.k)
    torch.cuda.synchronize(); t0 = time.time()
    r = E.sweep(gm, gk, st, nst, thin=1, seed=1215, want_bits=False, check=False)
    torch.cuda.synchronize(); wall = (time.time() - t0) / nst * 1e6
print("kernel:", abi.last_kernel())
allw = os.environ.get("STAMP_ALL", "0") == "1"      # the build with -DFMCMC_STAMP_WAVE=-1: every wave's stamps
raw = r.samples.cpu().numpy()[::4, 0, :]
d = raw[:, :16] / (nst - 1)                         # wave 0 = first chain of every workgroup; ticks per step
tps = np.median(d.sum(axis=1)) / wall
print("wall %.1f us per step (stamped build), %.0f ticks per us" % (wall, tps))
if allw:
    print("observations / barrier 2, us per step, median over workgroups, waves 0..7:")
    for w in range(8):
        dw = raw[:, 16 * w:16 * w + 16] / (nst - 1) / tps
        print("  wave %d (SIMD %d): observations %6.2f  barrier 1 %5.2f  barrier 2 %6.2f  sum %6.2f" % (w, w % 4, np.median(dw[:, 5]), np.median(dw[:, 4]), np.median(dw[:, 6]), np.median(dw.sum(axis=1))))
d = d / tps
names = ["rng tile", "proposal (A)", "sync", "publish", "barrier 1", "observations", "barrier 2", "gather+wave sum", "sync",
         "B: rest", "accept/store (C): rest + tail", "B: -", "B: -", "B: -", "B: -", "C: logpost"]
print("us per step: median / min / max over workgroups")
tot = 0.0
for i, nm in enumerate(names):
    col = d[:, i]
    print("%-26s %7.2f %7.2f %7.2f" % (nm, np.median(col), col.min(), col.max()))
    tot += np.median(col)
print("%-26s %7.2f" % ("sum of medians", tot))
