"""Diagnostic: s_memtime shares of the phases of mh_sweep_spec's register owner (build with -DSPEC_STAMP: lib/libfmcmc_amd_specstamp.so),
kernel_ram unbounded against bounded at C3's shape.  Not a benchmark.
   python tools/stamp_spec_owner.py [chains=1024] [n=10000]
phases: 0 wait for the total | 1 f1 | 2 adaptation + second slot + decision | 3,4 (adapt: covariance, factor) | 5 proposal + publish | 6 row stores | 7 prepare"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "fmcmc_amd", "lib", "libfmcmc_amd_specstamp.so")
os.environ["FMCMC_AMD_LIB"] = lib          # (read when fmcmc_amd is first imported)
from fmcmc_amd import build
if not os.path.exists(lib) or os.environ.get("REBUILD"):
    build.build(out=lib, extra_flags=["-DSPEC_STAMP"])
import numpy as np, torch
from fmcmc_amd import engine as E, _abi as abi
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
p, k, big = 3, 5, E.DBL_MAX
rng = np.random.default_rng(3)
X = rng.standard_normal((n, p)); y = 1.0 + X @ np.array([0.5, -0.5, 0.25]) + rng.standard_normal(n)
gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
init = np.array([1.0, 0.5, -0.5, 0.25, 1.0])[None, :] + 0.01 * rng.standard_normal((C, k))
nsteps = 3000
for name, lb in (("kernel_ram()", np.full(k, -big)), ("kernel_ram(lb = sigma > 0)", np.array([-big] * (k - 1) + [0.0]))):
    gk = E.KernelSpec(abi.KERNEL_RAM, k, np.zeros(k), np.ones(k), lb, np.full(k, big), np.zeros(k, np.uint8))
    st = E.ChainState(init, gk.kf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = E.sweep(gm, gk, st, nsteps, seed=5, want_draws=True, want_bits=False, check=False)
    e1.record(); torch.cuda.synchronize()
    d = r.draws.reshape(-1)[: C * 16].cpu().numpy().reshape(C, 16)
    per = np.median(d[:, :8] / nsteps, axis=0)
    print("%-28s %-10s %.2f us/step; ticks per step by phase: %s | sum %.0f" % (name, abi.last_kernel(), e0.elapsed_time(e1) * 1e3 / nsteps,
          " ".join("%6.0f" % t for t in per), per.sum()), flush=True)
