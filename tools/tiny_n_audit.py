"""Round 5: which kernel, and at what price, at TINY data (n = 50 .. 512 -- the sizes of the reference's README and vignettes) over the
number of covariates, the proposal kernel and the chain count.   python tools/tiny_n_audit.py [out.md]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi
big = E.DBL_MAX
rows = ["| family | n | p | kernel_* | chains | engine kernel | us / step |", "|---|---|---|---|---|---|---|"]
rng = np.random.default_rng(11)
for fam in ("linreg", "logistic"):
    for n in (50, 200, 512):
        for p in ((0, 1, 3, 7, 8, 12, 15, 20) if fam == "linreg" else (1, 4, 7, 8, 12)):
            if p:
                X = rng.standard_normal((n, p))
            if fam == "linreg":
                y = 1.0 + (X @ np.linspace(1, -1, p) if p else 0.0) + 2 * rng.standard_normal(n)
                gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y) if p else E.DeviceModel(abi.FAM_IID_NORMAL, None, y)
                k = p + 2
                base = np.array([0.0] * (p + 1) + [y.std()])
            else:
                y = (rng.uniform(size=n) < 1 / (1 + np.exp(-(0.2 + X @ np.linspace(.5, -.5, p))))).astype(np.float64)
                gm = E.DeviceModel(abi.FAM_LOGISTIC, X, y, intercept=True, guard=False, prior_div=8.0)
                k = p + 1
                base = np.zeros(k)
            for kind, name in ((abi.KERNEL_NORMAL, "normal"), (abi.KERNEL_ADAPT, "adapt"), (abi.KERNEL_RAM, "ram")):
                gk = E.KernelSpec(kind, k, np.zeros(k), np.full(k, .02), np.full(k, -big), np.full(k, big), np.zeros(k, np.uint8), **({"warmup": 100} if kind == abi.KERNEL_ADAPT else {}))
                for C in (4, 1024):
                    init = base[None, :] + 0.02 * rng.standard_normal((C, k))
                    if fam == "linreg":
                        init[:, -1] = np.abs(init[:, -1]) + 0.5
                    best = 1e9
                    steps = 1500
                    try:
                        for _ in range(2):
                            st = E.ChainState(init, k)
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record(); E.sweep(gm, gk, st, steps, want_bits=False, check=False); e1.record(); torch.cuda.synchronize()
                            best = min(best, e0.elapsed_time(e1) * 1e3 / steps)
                        rows.append("| %s | %d | %d | %s | %d | %s | %.2f |" % (fam, n, p, name, C, abi.last_kernel(), best))
                    except Exception as e:
                        rows.append("| %s | %d | %d | %s | %d | refused: %s | - |" % (fam, n, p, name, C, str(e)[:50]))
                    print(rows[-1], flush=True)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(rows) + "\n")
