import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi
big = E.DBL_MAX
K, CH, n, nst = 50, 512, 10000, 1000
rng = np.random.default_rng(20260104)
X4 = rng.standard_normal((n, K - 2)); b4 = rng.standard_normal(K - 1); y4 = b4[0] + X4 @ b4[1:] + 2 * rng.standard_normal(n)
init4 = np.concatenate([b4, [2.0]])[None, :] + 0.01 * rng.standard_normal((CH, K)); init4[:, -1] = np.abs(init4[:, -1])
z, o = np.zeros(K), np.ones(K)
gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X4, y4)
gk = E.KernelSpec(4, K, z, o, -big * o, big * o, np.zeros(K, np.uint8))
for bits in (0, 2048, 4096, 2048 + 4096):
    os.environ["FMCMC_AMD_DEBUG"] = "mode=%d" % bits
    best = 1e9
    for _ in range(3):
        st = E.ChainState(init4, gk.kf)
        torch.cuda.synchronize(); t0 = time.time()
        r = E.sweep(gm, gk, st, nst, seed=1215, want_bits=False, check=False)
        torch.cuda.synchronize(); best = min(best, (time.time() - t0) / nst * 1e6)
    print("ablation bits %5d: %.2f us per step" % (bits, best), abi.last_kernel())
