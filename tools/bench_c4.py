"""Config C4 on one GPU (kernel_ram, k = 50, n = 10,000, 512 chains): time per step, for A/B runs via FMCMC_AMD_LIB."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi
big = E.DBL_MAX
n, nst = 10000, int(sys.argv[1]) if len(sys.argv) > 1 else 400
CH = int(sys.argv[2]) if len(sys.argv) > 2 else 512
K = int(sys.argv[3]) if len(sys.argv) > 3 else 50          # parameters: intercept + (K - 2) covariates + sigma
rng = np.random.default_rng(20260104)
X4 = rng.standard_normal((n, K - 2)); b4 = rng.standard_normal(K - 1); y4 = b4[0] + X4 @ b4[1:] + 2 * rng.standard_normal(n)
init4 = np.concatenate([b4, [2.0]])[None, :] + 0.01 * rng.standard_normal((CH, K)); init4[:, -1] = np.abs(init4[:, -1])
z, o = np.zeros(K), np.ones(K)
gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X4, y4)
for kind, name in ((4, "kernel_ram"), (1, "kernel_normal")):
    gk = E.KernelSpec(kind, K, z, o * (0.002 if kind == 1 else 1.0), -big * o, big * o, np.zeros(K, np.uint8))
    best = 1e9
    for _ in range(3):
        st = E.ChainState(init4, gk.kf)
        torch.cuda.synchronize(); t = time.time()
        r = E.sweep(gm, gk, st, nst, seed=1215, want_bits=False, check=False)
        torch.cuda.synchronize(); best = min(best, time.time() - t)
    print(("%s k=" + str(K) + " n=10k C=%d: %.1f us per step, %.3e samples/s, checksum %016x") % (
        name, CH, best / nst * 1e6, CH * (nst - 1) / best, int(r.samples.view(torch.int64).sum().item()) & (2**64 - 1)))
