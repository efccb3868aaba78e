#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point (what the R shim calls): config C2's shape through
fmcmc_mcmc_run_host with ordinary (pageable) numpy buffers -- upload of the data, the sweep, 901 MB of kept rows back.

  python tools/bench_host.py [--kind normal|adapt|ram] [--chains 1024] [--iters 10000] [--n 10000] [--reps 3]

Prints one JSON line: wall time of the call (best of reps), MH samples/s, the bytes that came back.  DESIGN.md quotes it next to
the device-resident `value` of bench.py; it is never `value`."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import _abi as abi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="normal")
    ap.add_argument("--chains", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=10000)
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    rng = np.random.default_rng(20260102)
    n, p, k, Cn, nsteps = a.n, 3, 5, a.chains, a.iters
    X = rng.standard_normal((n, p))
    beta = np.array([1.0, -0.5, 0.25, 2.0])
    y = beta[0] + X @ beta[1:] + 1.5 * rng.standard_normal(n)
    Xc = np.ascontiguousarray(X.T)
    init = np.array([0, 0, 0, 0, float(np.std(y))])[None, :] + 0.1 * rng.standard_normal((Cn, k))
    init[:, -1] = np.abs(init[:, -1])
    kind = {"normal": abi.KERNEL_NORMAL, "adapt": abi.KERNEL_ADAPT, "ram": abi.KERNEL_RAM}[a.kind]
    mu = np.zeros(k); scale = np.full(k, 0.02); lb = np.full(k, -np.finfo(np.float64).max); ub = -lb; fixed = np.zeros(k, np.uint8)
    S = nsteps
    P = lambda arr: arr.ctypes.data
    best = None
    for rep in range(a.reps):
        th = np.ascontiguousarray(init.copy()); f0 = np.zeros(Cn); abs_iter = np.zeros(Cn, np.int64)
        Sig = np.zeros((Cn, k, k)); mp = np.zeros((Cn, k)); hm = np.zeros(Cn, np.int32); ne = np.zeros(Cn, np.int32)
        samples = np.empty((Cn, k, S)); lp = np.empty((Cn, S)); dr = np.empty((Cn, k, S))
        samples[:] = 0; lp[:] = 0; dr[:] = 0          # (touch the pages: a first-touch fault per page is not what is measured)
        acc = np.zeros(Cn, np.int64); bits = np.zeros((Cn, (nsteps + 31) // 32), np.uint32)
        status = np.zeros(Cn, np.int32); sstep = np.zeros(Cn, np.int64); stheta = np.zeros((Cn, k))
        m = abi.Model(abi.FAM_GAUSSIAN_LINREG, p, n, P(Xc), P(y), 1, 1, 0.0)
        kk = abi.Kernel(kind, k, P(mu), P(scale), P(lb), P(ub), P(fixed), 0, 1, 500 if a.kind != "normal" else 0, 0, float("inf"),
                        1e-4, 0.234, 2.38 ** 2 / k)
        r = abi.Run(Cn, nsteps, 0, 1, 77, 0, 0, 0, 0, None, None)
        st = abi.State(P(th), P(f0), P(abs_iter), P(Sig), P(mp), P(hm), P(ne), 1, 0)
        out = abi.Out(P(samples), P(lp), P(dr), P(acc), P(bits), P(status), P(sstep), P(stheta))
        t0 = time.perf_counter()
        rc = abi.lib().fmcmc_mcmc_run_host(C.byref(m), C.byref(kk), C.byref(r), C.byref(st), C.byref(out), 0)
        dt = time.perf_counter() - t0
        if rc != 0:
            raise SystemExit("fmcmc_mcmc_run_host failed: %s" % abi.last_error())
        best = dt if best is None else min(best, dt)
        cks = float(samples[:, :, -1].sum()) + float(lp[:, -1].sum())
    nbytes = samples.nbytes + lp.nbytes + dr.nbytes
    print(json.dumps({"entry": "fmcmc_mcmc_run_host", "kind": a.kind, "chains": Cn, "iters": nsteps, "n": n, "wall_ms": round(best * 1e3, 2),
                      "samples_per_s": Cn * nsteps / best, "bytes_back": nbytes, "kernel": abi.last_kernel() if hasattr(abi, "last_kernel") else None,
                      "checksum": cks}))


if __name__ == "__main__":
    main()
