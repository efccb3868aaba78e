"""Times the BASELINE.json configs C2-C5 on ONE GPU (C4/C5 with their per-GPU chain share)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi
big = E.DBL_MAX


def timeit(name, gm, gk, init, nsteps, thin=1, reps=2):
    best = 1e9
    for _ in range(reps):
        st = E.ChainState(init, gk.kf)
        torch.cuda.synchronize(); t = time.time()
        r = E.sweep(gm, gk, st, nsteps, thin=thin, seed=1215, want_bits=False, check=False)
        torch.cuda.synchronize(); best = min(best, time.time() - t)
    C = init.shape[0]
    print("%-44s C=%5d nsteps=%5d: %8.1f ms  %.3e samples/s  acc %.3f  err chains %d" % (
        name, C, nsteps, best * 1e3, C * (nsteps - 1) / best, r.accept_count.double().mean().item() / (nsteps - 1),
        int((r.status != 0).sum().item())))


rng = np.random.default_rng(20260102)
n = 10000
X = rng.standard_normal((n, 3)); y = 3 + X @ np.array([2, -1, .5]) + 4 * rng.standard_normal(n)
init = np.array([0, 0, 0, 0, y.std()])[None, :] + 0.1 * rng.standard_normal((1024, 5)); init[:, 4] = np.abs(init[:, 4])
gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
z5, o5 = np.zeros(5), np.ones(5)
nst = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
timeit("C2 kernel_normal linreg n=10k k=5", gm, E.KernelSpec(1, 5, z5, o5 * .02, -big * o5, big * o5, np.zeros(5, np.uint8)), init, nst)
timeit("C3 kernel_adapt  linreg n=10k k=5", gm, E.KernelSpec(3, 5, z5, o5, -big * o5, big * o5, np.zeros(5, np.uint8), warmup=500), init, nst)
timeit("   kernel_ram    linreg n=10k k=5", gm, E.KernelSpec(4, 5, z5, o5, -big * o5, big * o5, np.zeros(5, np.uint8)), init, nst)
# C4: k = 50 RAM, 512 chains per GPU
rng = np.random.default_rng(20260104)
X4 = rng.standard_normal((n, 48)); b4 = rng.standard_normal(49); y4 = b4[0] + X4 @ b4[1:] + 2 * rng.standard_normal(n)
init4 = np.concatenate([b4, [2.0]])[None, :] + 0.01 * rng.standard_normal((512, 50)); init4[:, -1] = np.abs(init4[:, -1])
z, o = np.zeros(50), np.ones(50)
timeit("C4 kernel_ram linreg n=10k k=50 (512/GPU)", E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X4, y4),
       E.KernelSpec(4, 50, z, o, -big * o, big * o, np.zeros(50, np.uint8)), init4, max(200, nst // 10))
# C5: logistic n = 100k, k = 6, 1024 chains per GPU, thin 10
rng = np.random.default_rng(20260105)
n5 = 100000
X5 = rng.standard_normal((n5, 5)); b5 = np.array([-1, .5, -.5, .25, -.25, 1.0])
y5 = (rng.uniform(size=n5) < 1 / (1 + np.exp(-(b5[0] + X5 @ b5[1:])))).astype(np.float64)
init5 = b5[None, :] + 0.01 * rng.standard_normal((1024, 6))
z, o = np.zeros(6), np.ones(6)
timeit("C5 normal_reflective logistic n=100k k=6", E.DeviceModel(abi.FAM_LOGISTIC, X5, y5, intercept=True, guard=False, prior_div=8.0),
       E.KernelSpec(2, 6, z, o * .01, -5 * o, 5 * o, np.zeros(6, np.uint8)), init5, max(100, nst // 20), thin=10)
