"""Option audit: which kernel, and at what price, the OPTIONS of the kernel_* constructors select at a fixed shape (Gaussian linreg
n = 10,000, p = 3, 1024 chains -- config C2 / C3's shape -- and logistic n = 30,000, p = 5): bounds, fixed parameters, update
schemes, kernel_adapt's window / stride, kernel_ram's constr / qfun, the mirror kernels.  The table says what leaving the
defaults costs.   python tools/option_audit.py [out.md]   (on the GPU box)"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi  # noqa: E402

C = int(os.environ.get("OA_CHAINS", "1024"))      # (OA_CHAINS / OA_N: the same audit at another size, e.g. the README's: 2 chains, n = 1000)
rows = []


def timed(fn, reps=3):
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best


def run(label, gm, k, init, kind, **kw):
    big = E.DBL_MAX
    lb = kw.pop("lb", np.full(k, -big)); ub = kw.pop("ub", np.full(k, big))
    fixed = kw.pop("fixed", np.zeros(k, np.uint8))
    scale = kw.pop("scale", np.full(k, .02)); mu = kw.pop("mu", np.zeros(k))
    thin = kw.pop("thin", 1)
    try:
        gk = E.KernelSpec(kind, k, mu, scale, lb, ub, fixed, **kw)

        def make(steps):
            def go():
                st = E.ChainState(init, int((fixed == 0).sum()))
                E.sweep(gm, gk, st, steps, thin=thin, want_bits=False, check=False)
            return go
        g = make(40); g(); torch.cuda.synchronize()
        t40 = timed(g, 1)
        steps = int(min(4000, max(60, 40 * 0.05 / max(t40, 1e-5))))
        dt = timed(make(steps))
        rows.append((label, abi.last_kernel(), dt / (steps - 1) * 1e6))
    except Exception as e:
        rows.append((label, "refused: %s" % str(e)[:60], float("nan")))
    print(rows[-1], flush=True)


rng = np.random.default_rng(5)
n, p = int(os.environ.get("OA_N", "10000")), 3
X = rng.standard_normal((n, p)); y = 1.0 + X @ np.linspace(1, -1, p) + 4 * rng.standard_normal(n)
k = p + 2
init = np.array([0.0] * (p + 1) + [y.std()])[None, :] + 0.05 * rng.standard_normal((C, k)); init[:, -1] = np.abs(init[:, -1])
gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
big = E.DBL_MAX
lbs = np.array([-big, -big, -big, -big, 0.001]); ubs = np.full(k, big)
fx = np.zeros(k, np.uint8); fx[1] = 1
L = "linreg n=%g p=3, %d chains: " % (n, C) if (n, C) != (10000, 1024) else "linreg n=1e4 p=3: "
run(L + "kernel_normal()", gm, k, init, abi.KERNEL_NORMAL)
run(L + "kernel_normal(fixed = one)", gm, k, init, abi.KERNEL_NORMAL, fixed=fx)
run(L + "kernel_normal(scheme = 'ordered')", gm, k, init, abi.KERNEL_NORMAL, scheme=abi.SCHEME_ORDERED)
run(L + "kernel_normal(scheme = 'random')", gm, k, init, abi.KERNEL_NORMAL, scheme=abi.SCHEME_RANDOM)
run(L + "kernel_normal_reflective(lb = sigma > 0)", gm, k, init, abi.KERNEL_NORMAL_REFLECTIVE, lb=lbs, ub=ubs)
run(L + "kernel_unif()", gm, k, init, abi.KERNEL_UNIF, mu=np.full(k, -0.02), scale=np.full(k, 0.04))
run(L + "kernel_unif_reflective(lb)", gm, k, init, abi.KERNEL_UNIF_REFLECTIVE, mu=np.full(k, -0.02), scale=np.full(k, 0.04), lb=lbs, ub=ubs)
run(L + "kernel_adapt(warmup = 500)", gm, k, init, abi.KERNEL_ADAPT, warmup=500)
run(L + "kernel_adapt(lb = sigma > 0)", gm, k, init, abi.KERNEL_ADAPT, warmup=500, lb=lbs, ub=ubs)
run(L + "kernel_adapt(fixed = one)", gm, k, init, abi.KERNEL_ADAPT, warmup=500, fixed=fx)
run(L + "kernel_adapt(freq = 2)", gm, k, init, abi.KERNEL_ADAPT, warmup=500, freq=2)
run(L + "kernel_adapt(bw = 100)", gm, k, init, abi.KERNEL_ADAPT, warmup=500, bw=100)
run(L + "kernel_ram()", gm, k, init, abi.KERNEL_RAM)
run(L + "kernel_ram(lb = sigma > 0)", gm, k, init, abi.KERNEL_RAM, lb=lbs, ub=ubs)
run(L + "kernel_ram(fixed = one)", gm, k, init, abi.KERNEL_RAM, fixed=fx)
run(L + "kernel_ram(freq = 2)", gm, k, init, abi.KERNEL_RAM, freq=2)
run(L + "kernel_ram(qfun = rnorm)", gm, k, init, abi.KERNEL_RAM, ram_qfun=1)
run(L + "kernel_nmirror()", gm, k, init, abi.KERNEL_NMIRROR, mu=init[0].copy(), scale=np.full(k, 0.1), warmup=500, nadapt=5)
run(L + "kernel_umirror()", gm, k, init, abi.KERNEL_UMIRROR, mu=init[0].copy(), scale=np.full(k, 0.1), warmup=500, nadapt=5)

n, p = 30000, 5
X = rng.standard_normal((n, p)); beta = np.concatenate([[-1.0], np.linspace(.5, -.5, p)])
y = (rng.uniform(size=n) < 1 / (1 + np.exp(-(beta[0] + X @ beta[1:])))).astype(np.float64)
k = p + 1
init = beta[None, :] + 0.01 * rng.standard_normal((C, k))
gm = E.DeviceModel(abi.FAM_LOGISTIC, X, y, intercept=True, guard=False, prior_div=8.0)
fx = np.zeros(k, np.uint8); fx[1] = 1
L = "logistic n=3e4 p=5: "
run(L + "kernel_normal()", gm, k, init, abi.KERNEL_NORMAL, scale=np.full(k, .01), thin=10)
run(L + "kernel_normal_reflective(lb, ub)", gm, k, init, abi.KERNEL_NORMAL_REFLECTIVE, scale=np.full(k, .01), lb=np.full(k, -5.0), ub=np.full(k, 5.0), thin=10)
run(L + "kernel_normal(fixed = one)", gm, k, init, abi.KERNEL_NORMAL, scale=np.full(k, .01), fixed=fx, thin=10)
run(L + "kernel_unif()", gm, k, init, abi.KERNEL_UNIF, mu=np.full(k, -0.01), scale=np.full(k, 0.02), thin=10)
run(L + "kernel_adapt()", gm, k, init, abi.KERNEL_ADAPT, warmup=500, thin=10)
run(L + "kernel_ram()", gm, k, init, abi.KERNEL_RAM, thin=10)
run(L + "kernel_normal(scheme = 'ordered')", gm, k, init, abi.KERNEL_NORMAL, scale=np.full(k, .01), scheme=abi.SCHEME_ORDERED, thin=10)

out = ["| call | engine kernel | us / step |", "|---|---|---|"] + ["| %s | %s | %.2f |" % r for r in rows]
text = "\n".join(out) + "\n"
print(text)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(text)
