"""The shape map (profiles/r04_shape_map.md): what a user gets away from the BASELINE shapes.  1024 chains; Gaussian linreg p = 3
at n in {2k, 5k, 9k, 10k, 10.3k, 12k, 20k, 50k} x {kernel_normal, kernel_adapt, kernel_ram}, logistic p in {5, 7} at n in
{10k, 50k, 100k}: MH samples/s, the kernel the dispatcher picked, and the rate per algorithmic flop (SURVEY 8d:
n (2 p + 3) for linreg with p covariates, n (2 p + 8) for logistic) -- adjacent shapes should not differ by more than 1.5x there.
  python tools/shape_map.py [out.md]   (on the GPU box; HIP events around the sweep, best of 3)"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi  # noqa: E402

C = 1024
rows = []


def timed(fn, reps=3):
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best


def linreg(n, p, kind, name):
    rng = np.random.default_rng(n + p)
    X = rng.standard_normal((n, p)); y = 1.0 + X @ np.linspace(1, -1, p) + 4 * rng.standard_normal(n)
    k = p + 2
    init = np.array([0.0] * (p + 1) + [y.std()])[None, :] + 0.05 * rng.standard_normal((C, k)); init[:, -1] = np.abs(init[:, -1])
    gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
    big = E.DBL_MAX
    gk = E.KernelSpec(kind, k, np.zeros(k), np.full(k, .02), np.full(k, -big), np.full(k, big), np.zeros(k, np.uint8),
                      warmup=(500 if kind == abi.KERNEL_ADAPT else 0))
    steps = int(max(300, min(4000, 4e7 / n)))

    def go():
        st = E.ChainState(init, k)
        E.sweep(gm, gk, st, steps, want_bits=False, check=False)
    dt = timed(go)
    rows.append(("linreg p=%d" % p, n, name, abi.last_kernel(), C * (steps - 1) / dt, n * (2 * p + 3)))


def logistic(n, p):
    rng = np.random.default_rng(7 * n + p)
    X = rng.standard_normal((n, p)); beta = np.concatenate([[-1.0], np.linspace(.5, -.5, p)])
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-(beta[0] + X @ beta[1:])))).astype(np.float64)
    k = p + 1
    init = beta[None, :] + 0.01 * rng.standard_normal((C, k))
    gm = E.DeviceModel(abi.FAM_LOGISTIC, X, y, intercept=True, guard=False, prior_div=8.0)
    gk = E.KernelSpec(abi.KERNEL_NORMAL_REFLECTIVE, k, np.zeros(k), np.full(k, .01), np.full(k, -5.0), np.full(k, 5.0), np.zeros(k, np.uint8))
    steps = int(max(100, min(2000, 2e7 / n)))

    def go():
        st = E.ChainState(init, k)
        E.sweep(gm, gk, st, steps, thin=10, want_bits=False, check=False)
    dt = timed(go)
    rows.append(("logistic p=%d" % p, n, "normal_reflective", abi.last_kernel(), C * (steps - 1) / dt, n * (2 * p + 8)))


for n in (2000, 5000, 9000, 10000, 10300, 12000, 20000, 50000):
    for kind, name in ((abi.KERNEL_NORMAL, "normal"), (abi.KERNEL_ADAPT, "adapt"), (abi.KERNEL_RAM, "ram")):
        linreg(n, 3, kind, name)
for p in (5, 7):
    for n in (10000, 50000, 100000):
        logistic(n, p)
out = ["| model | n | kernel_* | engine kernel | MH samples/s | algorithmic TFLOP/s | frac of 78.6 |", "|---|---|---|---|---|---|---|"]
for m, n, kn, ek, v, fl in rows:
    out.append("| %s | %d | %s | %s | %.3e | %.1f | %.2f |" % (m, n, kn, ek, v, v * fl / 1e12, v * fl / 78.6e12))
text = "\n".join(out)
print(text)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(text + "\n")
