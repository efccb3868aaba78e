"""Behavioural properties the reference's own test-suite pins (inst/tinytest/*.R), run against the
oracle in both modes.  These are the properties, not values: the values are in test_oracle_golden."""
import ctypes as C

import numpy as np
import pytest

from conftest import synth_linreg


def test_same_seed_same_answer(O):
    """test-mcmc.R:107-137."""
    X, y = synth_linreg(300, 1, 1)
    m = O.Model(O.FAM_LINREG, X, y)
    k = O.Kernel(O.K_NORMAL, 3, scale=0.1)
    a = O.run(m, k, [[0, 0, 4.0]] * 2, nsteps=400, seed=42)
    b = O.run(m, k, [[0, 0, 4.0]] * 2, nsteps=400, seed=42)
    c = O.run(m, k, [[0, 0, 4.0]] * 2, nsteps=400, seed=43)
    assert np.array_equal(a.samples, b.samples) and not np.array_equal(a.samples, c.samples)
    assert not np.array_equal(a.samples[0], a.samples[1])  # chains differ (chain id is in the counter)


@pytest.mark.parametrize("mode", ["philox", "rmt"])
def test_fixed_parameters_never_move(O, mode):
    """test-mcmc.R:161, test-kernel_unif.R:45."""
    X, y = synth_linreg(200, 2, 2)
    m = O.Model(O.FAM_LINREG, X, y)
    k = O.Kernel(O.K_NORMAL, 4, scale=0.1, fixed=[False, True, False, False])
    kw = dict(rng_mode=O.RNG_RMT, math_mode=O.MATH_R, rng=O.RRng(5)) if mode == "rmt" else {}
    r = O.run(m, k, [0, 1.5, 0, 4.0], nsteps=300, **kw)
    assert np.all(r.samples[0][:, 1] == 1.5) and np.all(r.draws[0][:, 1] == 1.5)
    assert r.samples[0][:, 0].std() > 0


def test_ordered_scheme_alternates(O):
    """test-kernel_normal.R:72-98: exactly one free coordinate moves per proposal, cycling."""
    X, y = synth_linreg(200, 2, 3)
    m = O.Model(O.FAM_LINREG, X, y)
    k = O.Kernel(O.K_NORMAL, 4, scale=0.1, scheme="ordered", fixed=[False, True, False, False])
    r = O.run(m, k, [0, 1.0, 0, 4.0], nsteps=60)
    free = [0, 2, 3]
    state, draws = r.samples[0], r.draws[0]
    for i in range(2, 61):  # loop index i: proposal differs from the previous state in ONE coordinate
        moved = np.nonzero(draws[i - 1] != state[i - 2])[0]
        assert list(moved) == [free[(i - 1) % 3]]


@pytest.mark.parametrize("mode", ["philox", "rmt"])
def test_reflective_stays_inside_and_reaches_both_ends(O, mode):
    """test-kernel_normal.R:27-70."""
    D = np.random.default_rng(4).normal(0, 1, 50)
    m = O.Model(O.FAM_IID_NORMAL, None, D, guard=True)
    k = O.Kernel(O.K_NORMAL_REFLECTIVE, 2, scale=2.0, lb=[-1.0, 0.5], ub=[1.0, 2.0])
    kw = dict(rng_mode=O.RNG_RMT, math_mode=O.MATH_R, rng=O.RRng(9)) if mode == "rmt" else {}
    r = O.run(m, k, [0.0, 1.0], nsteps=3000, **kw)
    d = r.draws[0]
    assert d[:, 0].min() >= -1 and d[:, 0].max() <= 1 and d[:, 1].min() >= 0.5 and d[:, 1].max() <= 2
    assert d[:, 0].min() < -0.95 and d[:, 0].max() > 0.95


def test_reflect_formula_matches_repeated_folding(O):
    """R/kernel.R:450-493 is the closed form of folding the line onto [lb, ub]."""
    rng = np.random.default_rng(0)
    lb, ub = np.array([-1.0, 0.0, 2.0]), np.array([1.0, 0.5, 7.0])
    which = np.arange(3, dtype=np.int32)
    for _ in range(300):
        x0 = rng.uniform(-30, 30, 3)
        for mode in (O.MATH_R, O.MATH_CANON):
            x = x0.copy()
            O.lib().fmcmc_oracle_reflect(O._p(x), O._p(lb), O._p(ub), which.ctypes.data_as(C.POINTER(C.c_int32)), 3, mode)
            ref = x0.copy()
            for j in range(3):
                while ref[j] > ub[j] or ref[j] < lb[j]:
                    ref[j] = 2 * ub[j] - ref[j] if ref[j] > ub[j] else 2 * lb[j] - ref[j]
            assert np.allclose(x, ref, atol=1e-12)


def test_na_bounds_equal_machine_limits(O):
    """test-na-bounds.R:66-93: NA bounds == +-.Machine$double.xmax, bit-identically."""
    X, y = synth_linreg(150, 1, 6)
    m = O.Model(O.FAM_LINREG, X, y)
    a = O.run(m, O.Kernel(O.K_NORMAL_REFLECTIVE, 3, scale=0.2, lb=[np.nan, np.nan, 0.0], ub=np.nan), [0, 0, 4.0],
              nsteps=200, seed=3)
    b = O.run(m, O.Kernel(O.K_NORMAL_REFLECTIVE, 3, scale=0.2, lb=[-O.DBL_MAX, -O.DBL_MAX, 0.0], ub=O.DBL_MAX),
              [0, 0, 4.0], nsteps=200, seed=3)
    assert np.array_equal(a.samples, b.samples)


def test_recursive_mean_cov_identity(O):
    """test-kernel_adapt.R:32-55: feeding rows one by one reproduces colMeans / cov to 1e-10."""
    rng = np.random.default_rng(1)
    Xm = rng.standard_normal((40, 4))
    k = 4
    mean = Xm[:2].mean(0).copy()
    cov = np.cov(Xm[:2].T).copy()
    Ik = np.zeros((k, k))
    for t in range(2, 40):
        x = np.ascontiguousarray(Xm[t])
        mnew = np.empty(k)
        O.lib().fmcmc_oracle_mean_recursive(O._p(x), O._p(mean), float(t), k, O._p(mnew))
        c = np.ascontiguousarray(cov)
        O.lib().fmcmc_oracle_cov_recursive(O._p(x), O._p(c), O._p(mean), O._p(mnew), float(t), 0.0, 1.0, O._p(Ik), k)
        mean, cov = mnew, c
        assert np.allclose(mean, Xm[:t + 1].mean(0), atol=1e-10)
        assert np.allclose(cov, np.cov(Xm[:t + 1].T), atol=1e-10)


def test_argument_errors(O):
    """test-mcmc.R:3-23, test-kernels.R:14-31."""
    X, y = synth_linreg(50, 1, 1)
    m = O.Model(O.FAM_LINREG, X, y)
    k = O.Kernel(O.K_NORMAL, 3)
    with pytest.raises(ValueError, match="burnin"):
        O.run(m, k, [0, 0, 1.0], nsteps=10, burnin=10)
    with pytest.raises(ValueError, match="thin"):
        O.run(m, k, [0, 0, 1.0], nsteps=10, thin=10)
    with pytest.raises(ValueError, match="thin"):
        O.run(m, k, [0, 0, 1.0], nsteps=10, thin=0)
    with pytest.raises(ValueError, match="-ub- cannot be <= than -lb-"):
        O.Kernel(O.K_NORMAL_REFLECTIVE, 3, lb=1.0, ub=1.0)
    with pytest.raises(ValueError, match="cannot be zero"):
        O.Kernel(O.K_NORMAL, 3, fixed=True)
    with pytest.raises(ValueError, match="Incorrect length of"):
        O.Kernel(O.K_NORMAL, 3, scale=[1.0, 2.0])


def test_nan_logposterior_is_an_error_with_step_and_theta(O):
    """R/mcmc.R:758-765 (README ll without the is.finite guard, sigma < 0)."""
    X, y = synth_linreg(100, 1, 2)
    m = O.Model(O.FAM_LINREG, X, y, guard=False)
    r = O.run(m, O.Kernel(O.K_NORMAL, 3, scale=1.0), [0, 0, 0.05], nsteps=200, seed=1)
    assert r.rc == 3 and r.status[0] == 1 and r.status_step[0] >= 2 and r.status_theta[0, 2] < 0
    g = O.run(O.Model(O.FAM_LINREG, X, y, guard=True), O.Kernel(O.K_NORMAL, 3, scale=1.0), [0, 0, 0.05], nsteps=200, seed=1)
    assert g.rc == 0 and g.samples[0][:, 2].min() > 0  # guarded: -Inf rejects instead


def test_burnin_thin_rows_and_labels(O):
    """R/mcmc.R:786-836, test-append_chains.R:10-60."""
    X, y = synth_linreg(100, 1, 3)
    m = O.Model(O.FAM_LINREG, X, y)
    k = O.Kernel(O.K_NORMAL, 3, scale=0.1)
    full = O.run(m, k, [0, 0, 4.0], nsteps=100, seed=9)
    thinned = O.run(m, k, [0, 0, 4.0], nsteps=100, burnin=20, thin=7, seed=9)
    assert list(thinned.iters) == list(range(27, 100, 7))
    assert np.array_equal(thinned.samples[0], full.samples[0][thinned.iters - 1])
    assert np.array_equal(thinned.logpost[0], full.logpost[0][thinned.iters - 1])


@pytest.mark.parametrize("kind", ["normal", "adapt", "ram"])
def test_posterior_recovery(O, kind):
    """test-mcmc.R:27-72,139-193; test-kernel_adapt.R:24-25; test-kernel_ram.R:26-27 (loose tolerances)."""
    rng = np.random.default_rng(1231)
    D = rng.normal(2.6, 3.0, 1000)
    m = O.Model(O.FAM_IID_NORMAL, None, D, guard=True)
    kern = {"normal": O.Kernel(O.K_NORMAL_REFLECTIVE, 2, scale=0.1, lb=[-10, 0.0], ub=10.0),
            "adapt": O.Kernel(O.K_ADAPT, 2, lb=[-10, 0.0], ub=10.0, warmup=200),
            "ram": O.Kernel(O.K_RAM, 2, lb=[-10, 0.0], ub=10.0)}[kind]
    r = O.run(m, kern, [[1.0, 1.0], [3.0, 4.0]], nsteps=6000, burnin=2000, seed=11)
    est = r.samples.reshape(-1, 2).mean(0)
    assert abs(est[0] - D.mean()) < 0.1 and abs(est[1] - D.std(ddof=1)) < 0.1
    if kind == "ram":
        acc = r.accept_count / 5999
        # adapts towards Vihola's 0.234 (README.md:245-246 reports 0.352 after 5000 steps of this kernel)
        assert np.all((acc > 0.15) & (acc < 0.5))


def test_gelman_matches_textbook_formula(O):
    rng = np.random.default_rng(5)
    m_, N, p = 4, 500, 3
    x = rng.standard_normal((m_, N, p)) + rng.standard_normal((m_, 1, p)) * 0.3
    psrf, mpsrf = O.gelman(x)
    W = np.mean([np.cov(c.T) for c in x], axis=0)
    B = N * np.cov(x.mean(1).T)
    lam = np.max(np.linalg.eigvals(np.linalg.solve(W, B)).real)
    assert abs(mpsrf - np.sqrt((1 - 1 / N) + (1 + 1 / p) * lam / N)) < 1e-10
    # univariate branch, vectorised numpy restatement of the same published formula
    w, b = np.diag(W), np.diag(B)
    s2 = np.array([np.var(c, axis=0, ddof=1) for c in x])
    xb = x.mean(1)
    mu = xb.mean(0)
    cov = lambda a, c: ((a - a.mean(0)) * (c - c.mean(0))).sum(0) / (m_ - 1)
    var_w, var_b = np.var(s2, axis=0, ddof=1) / m_, 2 * b ** 2 / (m_ - 1)
    cov_wb = (N / m_) * (cov(s2, xb ** 2) - 2 * mu * cov(s2, xb))
    V = (N - 1) * w / N + (1 + 1 / m_) * b / N
    var_V = ((N - 1) ** 2 * var_w + (1 + 1 / m_) ** 2 * var_b + 2 * (N - 1) * (1 + 1 / m_) * cov_wb) / N ** 2
    df_V = 2 * V ** 2 / var_V
    ref = np.sqrt((df_V + 3) / (df_V + 1) * ((N - 1) / N + (1 + 1 / m_) * b / w / N))
    assert np.allclose(psrf, ref, rtol=1e-10)


def test_philox_fixture_is_stable(O):
    """tests/golden/philox_vectors.json (made by make_philox_vectors.py from this oracle): guards drift."""
    import json, os
    from golden.make_philox_vectors import cases, run_case
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "philox_vectors.json")))
    for name, spec in cases().items():
        got = run_case(O, spec)
        for key, val in fx[name].items():
            assert got[key] == val, (name, key)
