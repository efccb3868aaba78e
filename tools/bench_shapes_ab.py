"""Same-box A/B of two library builds over Gaussian linreg shapes (normal and reflective kernels):
   python tools/bench_shapes_ab.py libA.so libB.so [nsteps]   (each shape runs in a child process per library)"""
import os, sys, subprocess, json
CHILD = r'''
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from fmcmc_amd import engine as E, _abi as abi
C, nsteps = 1024, int(sys.argv[1])
out = {}
for n, p, kind in [(10000, 3, 1), (10000, 3, 2), (8000, 3, 1), (5000, 3, 1), (2000, 3, 1), (600, 3, 1), (10000, 1, 1), (10000, 2, 1), (5000, 5, 1), (3000, 7, 1), (5000, 7, 2)]:
    rng = np.random.default_rng(n + p)
    X = rng.standard_normal((n, p)); y = 1.0 + X @ np.linspace(1, -1, p) + 4 * rng.standard_normal(n)
    k = p + 2
    init = np.array([0.0] * (p + 1) + [y.std()])[None, :] + 0.05 * rng.standard_normal((C, k)); init[:, -1] = np.abs(init[:, -1])
    gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
    lb = np.full(k, -E.DBL_MAX); ub = np.full(k, E.DBL_MAX)
    if kind == 2: lb = np.full(k, -10.0); ub = np.full(k, 10.0)
    gk = E.KernelSpec(kind, k, np.zeros(k), np.full(k, .02), lb, ub, np.zeros(k, np.uint8))
    best = 0.0
    for rep in range(3):
        st = E.ChainState(init, k)
        torch.cuda.synchronize(); t = time.time()
        r = E.sweep(gm, gk, st, nsteps, want_draws=True, want_logpost=True, want_bits=False)
        torch.cuda.synchronize(); best = max(best, C * (nsteps - 1) / (time.time() - t))
    out["n=%d p=%d kind=%d" % (n, p, kind)] = best
print("RESULT " + json.dumps(out))
'''
libs = sys.argv[1:3]
nsteps = sys.argv[3] if len(sys.argv) > 3 else "3000"
res = []
for L in libs:
    env = dict(os.environ, FMCMC_AMD_LIB=os.path.abspath(L))
    o = subprocess.run([sys.executable, "-c", CHILD, nsteps], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in o.stdout.splitlines() if l.startswith("RESULT ")]
    if not line:
        print(o.stdout[-2000:], o.stderr[-2000:]); sys.exit(1)
    res.append(json.loads(line[0][7:]))
for key in res[0]:
    a, b = res[0][key], res[1][key]
    print("%-24s %s %.3e | %s %.3e | x%.3f" % (key, os.path.basename(libs[0]), a, os.path.basename(libs[1]), b, b / a))
