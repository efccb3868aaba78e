"""Generates tests/golden/philox_vectors.json: small expected outputs of the engine's canonical
(PHILOX) stream, produced by the CPU oracle.  The GPU tests compare the HIP engine with these vectors
as well as with the live oracle; a CPU test re-runs this script's cases to catch oracle drift.
Inputs are generated with numpy's default_rng (seeded); outputs are stored as hex of the fp64 bits.

    python tests/golden/make_philox_vectors.py      # rewrites the fixture
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def cases():
    return {
        "normal_linreg": dict(kind="normal", n=700, p=2, C=3, nsteps=120, seed=1215, data_seed=101, scale=0.05),
        "reflective_linreg": dict(kind="reflective", n=700, p=1, C=2, nsteps=150, seed=7, data_seed=102, scale=0.4),
        "adapt_linreg": dict(kind="adapt", n=600, p=2, C=2, nsteps=160, seed=8, data_seed=103, warmup=40),
        "ram_linreg": dict(kind="ram", n=600, p=2, C=2, nsteps=120, seed=9, data_seed=104),
        "headline_shape": dict(kind="normal", n=10000, p=3, C=2, nsteps=40, seed=1215, data_seed=20260102, scale=0.02),
        "unif_random_scheme": dict(kind="unif_reflective", n=500, p=2, C=3, nsteps=140, seed=21, data_seed=105, scheme="random"),
        "normal_explicit_scheme": dict(kind="normal", n=500, p=2, C=2, nsteps=140, seed=22, data_seed=106, scale=0.08,
                                       scheme=[4, 2, 1, 3]),
        "ram_freq_constr": dict(kind="ram", n=500, p=2, C=2, nsteps=140, seed=23, data_seed=107, freq=3, constr="tridiag"),
    }


def make_inputs(spec):
    rng = np.random.default_rng(spec["data_seed"])
    n, p, C = spec["n"], spec["p"], spec["C"]
    X = rng.standard_normal((n, p))
    beta = np.array([3.0, 2.0, -1.0, 0.5])[: p + 1]
    y = beta[0] + X @ beta[1:] + 4.0 * rng.standard_normal(n)
    init = np.array([0.0] * (p + 1) + [float(np.std(y))])[None, :] + 0.1 * rng.standard_normal((C, p + 2))
    init[:, -1] = np.abs(init[:, -1])
    return X, y, init


def kernel_kwargs(O, spec):
    k = spec["p"] + 2
    if spec["kind"] == "normal":
        return O.K_NORMAL, dict(scale=spec["scale"], scheme=spec.get("scheme", "joint"))
    if spec["kind"] == "unif_reflective":
        return O.K_UNIF_REFLECTIVE, dict(min_=-0.1, max_=0.15, lb=[-9.0] * (k - 1) + [0.2], ub=9.0,
                                         scheme=spec.get("scheme", "joint"))
    if spec["kind"] == "ram" and "constr" in spec:
        M = (np.abs(np.subtract.outer(np.arange(k), np.arange(k))) <= 1).astype(float)
        return O.K_RAM, dict(freq=spec["freq"], constr=M)
    if spec["kind"] == "reflective":
        return O.K_NORMAL_REFLECTIVE, dict(scale=spec["scale"], lb=[-5.0] * (k - 1) + [0.1], ub=[5.0] * k)
    if spec["kind"] == "adapt":
        return O.K_ADAPT, dict(warmup=spec["warmup"])
    return O.K_RAM, {}


def hexbits(a):
    return [format(int(v), "016x") for v in np.ascontiguousarray(a, dtype=np.float64).view(np.uint64).ravel()]


def run_case(O, spec):
    X, y, init = make_inputs(spec)
    kind, kw = kernel_kwargs(O, spec)
    r = O.run(O.Model(O.FAM_LINREG, X, y), O.Kernel(kind, spec["p"] + 2, **kw), init, nsteps=spec["nsteps"],
              seed=spec["seed"])
    return {"accept_count": [int(v) for v in r.accept_count],
            "accept_bits": [int(v) for v in r.accept_bits.ravel()],
            "last_row": hexbits(r.samples[:, -1, :]),
            "logpost_last": hexbits(r.logpost[:, -1]),
            "sample_checksum": hexbits([r.samples.sum(), np.abs(r.draws).sum()])}


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    out = {name: run_case(O, spec) for name, spec in cases().items()}
    json.dump(out, open(os.path.join(HERE, "philox_vectors.json"), "w"), indent=1)
    print("wrote", len(out), "cases")
