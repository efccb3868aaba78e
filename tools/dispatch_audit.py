"""Dispatch audit (profiles/r04_dispatch_audit.md): does the dispatcher pick the fastest kernel it has?  For a grid of shapes
(Gaussian linreg p = 1 .. 14 and wide, logistic) x chain counts x kernel_* every sweep is timed on the dispatcher's own choice and
on the alternatives the diagnosis knobs can force (FMCMC_AMD_DEBUG, read once per call: mfma=0, lat=0|1|2|3, streamed=1, shard=0|1,
wide2=0, shard_mfma=0); a row is flagged when an alternative beats the default by more than 5 %.
  python tools/dispatch_audit.py [out.md] [--quick] [--only=narrow,few,wide,logistic,long]      (on the GPU box; HIP events around the sweep, best of 3)"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import engine as E, _abi as abi  # noqa: E402

QUICK = "--quick" in sys.argv
ONLY = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]
ONLY = ONLY[0] if ONLY else ["narrow", "few", "wide", "logistic", "long"]      # --only=wide,logistic
rows = []
t_begin = time.time()


def timed(fn, reps=3):
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best


def measure(make_go, alts):
    """make_go(steps) -> callable; returns [(knobs, kernel, us per step)]"""
    out = []
    for knobs in alts:
        if knobs:
            os.environ["FMCMC_AMD_DEBUG"] = knobs
        else:
            os.environ.pop("FMCMC_AMD_DEBUG", None)
        try:
            go = make_go(40)
            go(); torch.cuda.synchronize()
            t40 = timed(go, 1)
            steps = int(min(6000, max(60, 40 * 0.05 / max(t40, 1e-5))))       # ~50 ms of sweep
            go = make_go(steps)
            dt = timed(go)
            out.append((knobs or "(default)", abi.last_kernel(), dt / (steps - 1) * 1e6))
        except Exception as e:   # an alternative the shape does not admit
            out.append((knobs, "refused: %s" % str(e)[:40], float("inf")))
    os.environ.pop("FMCMC_AMD_DEBUG", None)
    return out


def linreg(n, p, chains, kind, name, alts):
    rng = np.random.default_rng(n + p)
    X = rng.standard_normal((n, p)); y = 1.0 + X @ np.linspace(1, -1, p) + 4 * rng.standard_normal(n)
    k = p + 2
    init = np.array([0.0] * (p + 1) + [y.std()])[None, :] + 0.05 * rng.standard_normal((chains, k)); init[:, -1] = np.abs(init[:, -1])
    gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
    big = E.DBL_MAX
    gk = E.KernelSpec(kind, k, np.zeros(k), np.full(k, .02 / max(1.0, (p / 3.0) ** 0.5)), np.full(k, -big), np.full(k, big), np.zeros(k, np.uint8),
                      warmup=(500 if kind == abi.KERNEL_ADAPT else 0))

    def make_go(steps):
        def go():
            st = E.ChainState(init, k)
            E.sweep(gm, gk, st, steps, want_bits=False, check=False)
        return go
    rows.append(("linreg p=%d" % p, n, chains, name, measure(make_go, alts)))


def logistic(n, p, chains, alts):
    rng = np.random.default_rng(7 * n + p)
    X = rng.standard_normal((n, p)); beta = np.concatenate([[-1.0], np.linspace(.5, -.5, p)])
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-(beta[0] + X @ beta[1:])))).astype(np.float64)
    k = p + 1
    init = beta[None, :] + 0.01 * rng.standard_normal((chains, k))
    gm = E.DeviceModel(abi.FAM_LOGISTIC, X, y, intercept=True, guard=False, prior_div=8.0)
    gk = E.KernelSpec(abi.KERNEL_NORMAL_REFLECTIVE, k, np.zeros(k), np.full(k, .01), np.full(k, -5.0), np.full(k, 5.0), np.zeros(k, np.uint8))

    def make_go(steps):
        def go():
            st = E.ChainState(init, k)
            E.sweep(gm, gk, st, steps, thin=10, want_bits=False, check=False)
        return go
    rows.append(("logistic p=%d" % p, n, chains, "normal_reflective", measure(make_go, alts)))


KINDS = ((abi.KERNEL_NORMAL, "normal"), (abi.KERNEL_ADAPT, "adapt"), (abi.KERNEL_RAM, "ram"))
NARROW_ALTS = ["", "mfma=0", "streamed=1"]
# few chains per GPU (round 5: the latency form): the dispatcher's choice against the four-chains-per-workgroup kernels (lat=0) and
# against every forced number of chains per workgroup
FEW_ALTS = ["", "lat=0", "lat=1", "lat=2", "lat=3", "streamed=1"]
WIDE_ALTS = ["", "shard=0", "shard=1", "shard=1,wide2=0", "shard=1,shard_mfma=0"]
LOGIT_ALTS = ["", "shard=0", "shard=1"]
if QUICK:
    grid_narrow = [(n, p, c) for n in (1000, 10000, 15000) for p in (3,) for c in (1024,)]
    grid_wide = [(5000, 48, 512)]
    grid_logit = [(30000, 5, 1024)]
else:
    grid_narrow = [(n, p, c) for n in (300, 1000, 3000, 6000, 10000, 10241, 15000, 30000) for p in (1, 3, 5, 7, 10, 14) for c in (64, 1024, 4096)]
    grid_wide = [(n, p, c) for n in (1000, 5000, 10000, 20000) for p in (16, 30, 48, 60) for c in (64, 512, 2048)]
    grid_logit = [(n, p, c) for n in (2000, 10000, 30000, 100000) for p in (2, 5, 8, 12) for c in (64, 512, 1024, 4096)]
for n, p, c in (grid_narrow if "narrow" in ONLY else []):
    for kind, name in KINDS:
        linreg(n, p, c, kind, name, NARROW_ALTS)
    print("narrow", n, p, c, "%.0f s" % (time.time() - t_begin), flush=True)
grid_few = [] if QUICK else [(n, p, c) for n in (1000, 3000, 6000, 10000) for p in (1, 3, 5, 7) for c in (4, 128, 256, 512, 768)
                             if not (p >= 4 and n > 5120) and not (p >= 6 and n > 4096)]
for n, p, c in (grid_few if "few" in ONLY else []):
    for kind, name in KINDS:
        linreg(n, p, c, kind, name, FEW_ALTS)
    print("few", n, p, c, "%.0f s" % (time.time() - t_begin), flush=True)
for n, p, c in (grid_wide if "wide" in ONLY else []):
    for kind, name in (KINDS[0], KINDS[2]):
        linreg(n, p, c, kind, name, WIDE_ALTS)
    print("wide", n, p, c, "%.0f s" % (time.time() - t_begin), flush=True)
# few chains on long data (the long-data form, shard_long): default against shard=0 (the chain-sharded kernels) and shard=1 (forced)
grid_long = [] if QUICK else [(n, p, c) for n in (20000, 50000, 100000, 300000, 1000000) for p in (3, 7, 12) for c in (1, 4, 16, 64)]
for n, p, c in (grid_long if "long" in ONLY else []):
    for kind, name in KINDS:
        linreg(n, p, c, kind, name, ["", "shard=0", "shard=1"])
    print("long", n, p, c, "%.0f s" % (time.time() - t_begin), flush=True)
for n, p, c in ([] if QUICK or "long" not in ONLY else [(n, p, c) for n in (30000, 100000) for p in (30, 48) for c in (1, 4, 64)]):
    for kind, name in (KINDS[0], KINDS[2]):
        linreg(n, p, c, kind, name, ["", "shard=0", "shard=1"])
    print("long wide", n, p, c, "%.0f s" % (time.time() - t_begin), flush=True)
grid_llong = [] if QUICK else [(n, p, c) for n in (20000, 100000, 1000000) for p in (5, 12) for c in (1, 4, 16, 64)]
for n, p, c in (grid_llong if "long" in ONLY else []):
    logistic(n, p, c, ["", "shard=0", "shard=1"])
    print("long logistic", n, p, c, "%.0f s" % (time.time() - t_begin), flush=True)
for n, p, c in (grid_logit if "logistic" in ONLY else []):
    logistic(n, p, c, LOGIT_ALTS)
    print("logistic", n, p, c, "%.0f s" % (time.time() - t_begin), flush=True)

out = ["| model | n | chains | kernel_* | default kernel | us / step | fastest alternative (knobs: kernel) | us / step | default / fastest |", "|---|---|---|---|---|---|---|---|---|"]
flagged = 0
for m, n, c, kn, res in rows:
    d = res[0]
    alts = [r for r in res[1:] if r[1] != d[1] and np.isfinite(r[2])]
    best = min(alts, key=lambda r: r[2]) if alts else None
    ratio = d[2] / best[2] if best else float("nan")
    flag = " **<-**" if best and ratio > 1.05 else ""
    flagged += bool(flag)
    out.append("| %s | %d | %d | %s | %s | %.2f | %s | %s | %s%s |" % (m, n, c, kn, d[1], d[2], ("%s: %s" % (best[0], best[1])) if best else "(none differs)",
                                                                    ("%.2f" % best[2]) if best else "", ("%.2f" % ratio) if best else "", flag))
text = "\n".join(out) + "\n\n%d of %d rows flagged (an alternative more than 5 %% faster than the dispatcher's choice); %.0f s on the box.\n" % (flagged, len(rows), time.time() - t_begin)
print(text)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
if args:
    open(args[0], "w").write(text)
