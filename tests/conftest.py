import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def readme_data(O):
    """README.md:111-116: set.seed(78845); n <- 1000; X <- rnorm(n); y <- 3 + 2*X + rnorm(n, sd = 4)."""
    g = O.RRng(78845)
    X = g.rnorm(1000)
    y = 3.0 + 2.0 * X + g.rnorm(1000, 0.0, 4.0)
    return X, y


def synth_linreg(n, p, seed, beta=None, sigma=4.0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, p))
    if beta is None:
        beta = np.array([3.0, 2.0, -1.0, 0.5, 0.25, -0.75, 1.5, -2.0])[: p + 1]
    y = beta[0] + X @ beta[1:] + sigma * rng.standard_normal(n)
    return X, y


def set_knob(monkeypatch, key, value):
    """One diagnosis knob of the engine: merges `key=value` into FMCMC_AMD_DEBUG (the only variable the library reads)."""
    cur = dict(kv.split("=", 1) for kv in os.environ.get("FMCMC_AMD_DEBUG", "").split(",") if "=" in kv)
    cur[key] = str(value)
    monkeypatch.setenv("FMCMC_AMD_DEBUG", ",".join("%s=%s" % kv for kv in cur.items()))
