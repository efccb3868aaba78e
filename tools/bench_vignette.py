"""The reference's workflow vignette end to end through the host API (vignettes/workflow-with-fmcmc.Rmd): mcmc::logit-sized logistic
model (n = 100, k = 5), nchains = 2, kernel_adapt(freq = 1, warmup = 500), conv_checker = convergence_gelman(200), nsteps = 1e4 --
wall time of MCMC() per bulk of 200 steps (host logic + launches + the sweep)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings
warnings.filterwarnings('ignore')
import fmcmc_amd as fm
rng = np.random.default_rng(42)
n, p = 100, 4
X = rng.standard_normal((n, p)); beta = np.array([0.6, 0.8, 0.4, -0.5, 0.7])
y = (rng.uniform(size=n) < 1 / (1 + np.exp(-(beta[0] + X @ beta[1:])))).astype(np.float64)
fun = fm.logistic(X, y, intercept=True, prior_div=8.0)
for label, kw in (("kernel_normal, 1e4 steps, 2 chains", dict(kernel=fm.kernel_normal(scale=0.2), nsteps=10000, nchains=2)),
                  ("kernel_adapt, 1e4 steps, 2 chains", dict(kernel=fm.kernel_adapt(freq=1, warmup=500), nsteps=10000, nchains=2)),
                  ("kernel_adapt + convergence_gelman(200), up to 1e4 steps", dict(kernel=fm.kernel_adapt(freq=1, warmup=500), nsteps=10000, nchains=2,
                                                                                   conv_checker=fm.convergence_gelman(200))),
                  ("kernel_adapt + convergence_gelman(200), threshold 1.0 (never stops: 50 bulks)", dict(kernel=fm.kernel_adapt(freq=1, warmup=500), nsteps=10000, nchains=2,
                                                                                   conv_checker=fm.convergence_gelman(200, threshold=1.0)))):
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t = time.time()
        out = fm.MCMC(beta, fun, seed=1 + rep, **kw)
        torch.cuda.synchronize(); best = min(best, time.time() - t)
    first = out[0] if not hasattr(out, "data") else out
    rows = np.asarray(getattr(first, "data", first)).shape[0]
    print("%-80s %8.2f ms  (%d rows kept)" % (label, best * 1e3, rows), flush=True)
