"""Round 5: the calls that moved onto the wave-specialised kernel at the end of the round -- the bounded kernel_ram (knob specbnd), the mirror
kernels (specmirror), models without a covariate under the adaptive kernels (specp0) -- against their round-4 routes, over shapes and
chain counts: us per MH step with the knob at its default and at 0.  Each measurement in a child process (the knobs are read once).
   python tools/bench_spec_routes.py [out.md]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, %r)
from fmcmc_amd import engine as E, _abi as abi
what, C, n, p = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rng = np.random.default_rng(5)
big = E.DBL_MAX
if p:
    X = rng.standard_normal((n, p)); y = 1.0 + X @ np.linspace(1, -1, p) + 4 * rng.standard_normal(n)
    gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
else:
    y = 1.0 + 4 * rng.standard_normal(n)
    gm = E.DeviceModel(abi.FAM_IID_NORMAL, None, y)
k = p + 2
init = np.array([0.0] * (p + 1) + [y.std()])[None, :] + 0.05 * rng.standard_normal((C, k)); init[:, -1] = np.abs(init[:, -1])
lb, ub = np.full(k, -big), np.full(k, big)
if what == "ram_bounded":
    lb[-1] = 0.001
    gk = E.KernelSpec(abi.KERNEL_RAM, k, np.zeros(k), np.full(k, .02), lb, ub, np.zeros(k, np.uint8))
elif what == "nmirror":
    gk = E.KernelSpec(abi.KERNEL_NMIRROR, k, init[0].copy(), np.full(k, .02), lb, ub, np.zeros(k, np.uint8), warmup=500, nadapt=100)
else:
    gk = E.KernelSpec(abi.KERNEL_ADAPT, k, np.zeros(k), np.full(k, .02), lb, ub, np.zeros(k, np.uint8), warmup=100)
steps = 2000
best = 1e9
for _ in range(3):
    st = E.ChainState(init, k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); E.sweep(gm, gk, st, steps, want_bits=False, check=False); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3 / steps)
print("%%s %%.3f" %% (abi.last_kernel(), best))
''' % ROOT
rows = ["| call | chains | n | p | default: kernel, us / step | knob = 0: kernel, us / step | ratio |", "|---|---|---|---|---|---|---|"]
cases = [("ram_bounded", "specbnd", p, n) for p, n in ((1, 1000), (3, 10000), (3, 3000), (5, 5000), (7, 4000))] + \
        [("nmirror", "specmirror", p, n) for p, n in ((1, 1000), (3, 10000), (3, 3000), (5, 5000), (7, 4000))] + \
        [("adapt_p0", "specp0", 0, n) for n in (600, 3000, 10000)]
for what, knob, p, n in cases:
    for C in (256, 1024, 4096):
        res = []
        for val in (None, "0"):
            env = dict(os.environ)
            if val is not None:
                env["FMCMC_AMD_DEBUG"] = "%s=%s" % (knob, val)
            out = subprocess.run([sys.executable, "-c", CHILD, what, str(C), str(n), str(p)], env=env, capture_output=True, text=True)
            res.append(out.stdout.strip().split("\n")[-1] if out.returncode == 0 else "failed " + out.stderr.strip()[-80:])
        try:
            ratio = "%.2f" % (float(res[0].split()[-1]) / float(res[1].split()[-1]))
        except Exception:
            ratio = "-"
        rows.append("| %s | %d | %d | %d | %s | %s | %s |" % (what, C, n, p, res[0], res[1], ratio))
        print(rows[-1], flush=True)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(rows) + "\n")
