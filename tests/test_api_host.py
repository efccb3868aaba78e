"""Host-side logic of the drop-in API (no GPU): constructors, argument contracts and the bookkeeping
around the hot path, against the reference's documented behaviour."""
import warnings

import numpy as np
import pytest

import fmcmc_amd as f
from fmcmc_amd.convergence import _window
from fmcmc_amd.mcmc import Mcmc, McmcList


def test_kernel_constructors_expand_like_the_reference():
    k = f.kernel_normal_reflective(ub=5.0, lb=[-5.0, 0.0, 0.0], scale=0.05)
    assert k.k is None                       # lazily initialised on first use (R/kernel_normal.R:39)
    k._init(3)
    assert list(k.lb) == [-5, 0, 0] and list(k.ub) == [5, 5, 5] and list(k.scale) == [.05] * 3 and k.k == 3
    k2 = f.kernel_normal_reflective(lb=[np.nan, 0.0], ub=np.nan)      # R/kernel.R:25-41
    k2._init(2)
    assert k2.lb[0] == -np.finfo(float).max and k2.ub[1] == np.finfo(float).max
    ka = f.kernel_adapt()
    ka._init(5)
    assert ka.warmup == 500 and ka.eps == 1e-4 and abs(ka.Sd - 5.76 / 5) < 1e-15   # R/kernel_adapt.R:61,113-114
    kr = f.kernel_ram(fixed=[False, True, False])
    kr._init(3)
    assert kr.k == 2 and list(kr.which_) == [0, 2] and kr.arate == 0.234
    ko = f.kernel_normal(scheme="ordered")
    ko._init(4)
    assert ko.k == 1                          # k <<- sum(update_sequence[1,])


@pytest.mark.parametrize("make,msg", [
    (lambda: f.kernel_normal_reflective(lb=1.0, ub=1.0)._init(2), "-ub- cannot be <= than -lb-."),
    (lambda: f.kernel_normal(fixed=True)._init(3), "cannot be zero"),
    (lambda: f.kernel_normal(scale=[1.0, 2.0])._init(3), "Incorrect length of -scale-."),
    (lambda: f.kernel_adapt(bw=600, warmup=500), "The `warmup` parameter must be greater than `bw`."),
    (lambda: f.kernel_normal(scheme="sideways")._init(2), "-scheme- update must be"),
])
def test_kernel_errors(make, msg):
    """test-kernels.R:14-86."""
    with pytest.raises(ValueError, match=msg.replace("(", r"\(").replace(")", r"\)")):
        make()


def test_check_initial():
    """R/checks.R:22-58, test-checks.R:31-41."""
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        a, names = f.check_initial([1.0, 2.0], 3)
    assert a.shape == (3, 2) and names == ["par1", "par2"] and any("recycled" in str(x.message) for x in w)
    a, names = f.check_initial({"mu": 1.0, "sigma": 2.0}, 1)
    assert names == ["mu", "sigma"]
    with pytest.raises(ValueError, match="must coincide with the number of chains"):
        f.check_initial(np.zeros((2, 3)), 3)
    with pytest.raises(ValueError, match="length zero"):
        f.check_initial([], 1)


def test_mcmc_argument_checks_come_before_the_gpu():
    """test-mcmc.R:3-23: same messages, raised without touching a device."""
    fun = f.gaussian_linreg(np.arange(5.0), np.arange(5.0))
    for kw, msg in ((dict(burnin=10), "burnin"), (dict(thin=10), "thin"), (dict(thin=0), "-thin- should be >= 1"),
                    (dict(multicore=True), "`nchains` should be greater than 1"), (dict(nchains=0), "`nchains` must be")):
        with pytest.raises(ValueError, match=msg):
            f.MCMC([0, 0, 1.0], fun, 10, **kw)
    with pytest.raises(TypeError, match="closed-form families"):
        f.MCMC([0, 0, 1.0], lambda p: 0.0, 10)
    with pytest.raises(ValueError, match="Incorrect length of -initial-"):
        f.MCMC([0, 0, 0, 1.0], fun, 10)


def test_mcmc_containers_and_append_chains():
    """R/append_chains.R:90-143, test-append_chains.R:10-60."""
    a = Mcmc(np.arange(20.0).reshape(10, 2), start=1, end=10, thin=1)
    b = Mcmc(np.arange(20.0, 40.0).reshape(10, 2), start=1, end=10, thin=1)
    ab = f.append_chains(a, b)
    assert ab.niter == 20 and ab.mcpar == (1, 20, 1) and list(ab.iters) == list(range(1, 21))
    t1 = Mcmc(np.zeros((5, 1)), start=12, end=20, thin=2)
    t2 = Mcmc(np.ones((3, 1)), start=2, end=6, thin=2)
    t12 = f.append_chains(t1, t2)
    assert list(t12.iters) == [12, 14, 16, 18, 20, 22, 24, 26]
    with pytest.raises(ValueError, match="same `thin`"):
        f.append_chains(a, t1)
    la = f.append_chains(McmcList([a, a]), McmcList([b, b]))
    assert la.nchain == 2 and la[1].niter == 20
    assert a.tail(0).shape == (1, 2) and a.tail(2).shape == (3, 2)   # utils::tail as fmcmc uses it
    with pytest.raises(ValueError):
        Mcmc(np.zeros((4, 1)), start=1, end=10, thin=1)


def test_shard_bounds_partition_the_chains():
    for C in (1, 7, 1024, 4096):
        for W in (1, 2, 3, 8):
            b = [f.shard_bounds(C, W, r) for r in range(W)]
            assert b[0][0] == 0 and b[-1][1] == C and all(b[i][1] == b[i + 1][0] for i in range(W - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_gelman_window_is_codas_autoburnin():
    assert _window(np.arange(1, 201)) == (100, 100)        # keep iterations >= end/2 + 1 = 101
    assert _window(np.arange(1, 202)) == (101, 100)        # odd end: >= 101.5
    assert _window(np.arange(150, 201)) == (0, 51)          # start >= end/2: everything
    assert _window(np.arange(10, 1001, 10)) == (50, 50)     # thinned labels


def test_model_families_definitions():
    rng = np.random.default_rng(0)
    X = rng.standard_normal((50, 2)); y = rng.standard_normal(50)
    fun = f.gaussian_linreg(X, y)
    th = np.array([0.1, 0.2, -0.3, 1.5])
    from scipy.stats import norm
    assert abs(fun(th) - norm.logpdf(y - (th[0] + X @ th[1:3]), scale=th[3]).sum()) < 1e-9
    assert fun([0, 0, 0, -1.0]) == -np.inf and f.gaussian_linreg(X, y, guard=False)([0, 0, 0, 0.0]) == -np.inf
    assert np.isnan(f.gaussian_linreg(X, y, guard=False)([0, 0, 0, -1.0]))
    yb = (rng.uniform(size=50) < 0.5).astype(float)
    lg = f.logistic(np.column_stack([np.ones(50), X]), yb)
    b = np.array([0.3, -0.2, 0.1])
    eta = np.column_stack([np.ones(50), X]) @ b
    ref = np.sum(yb * (-np.log1p(np.exp(-eta))) + (1 - yb) * (-np.log1p(np.exp(eta)))) - np.sum(b ** 2) / 8
    assert abs(lg(b) - ref) < 1e-9 and lg.k == 3
    assert f.iid_normal(y).k == 2


def test_recursive_helpers_match_reference_identities_and_oracle(O):
    """test-kernel_adapt.R:32-55 (identity with colMeans / cov to 1e-10) + element-wise agreement with the oracle."""
    rng = np.random.default_rng(2)
    Xm = rng.standard_normal((30, 4))
    mean, cov = Xm[:2].mean(0), np.cov(Xm[:2].T)
    means = f.mean_recursive(Xm[2:], mean, 2)                      # matrix input: row-by-row recursion
    covs = f.cov_recursive(Xm[2:], cov, mean, 2, Mean_t=means)
    for i in range(28):
        assert np.allclose(means[i], Xm[:i + 3].mean(0), atol=1e-10)
        assert np.allclose(covs[i], np.cov(Xm[:i + 3].T), atol=1e-10)
    # single-row step against the oracle's C restatement (same operation order)
    x, mp, t = Xm[5], Xm[:5].mean(0), 5.0
    mt = f.mean_recursive(x, mp, t)
    mo = np.empty(4); O.lib().fmcmc_oracle_mean_recursive(O._p(np.ascontiguousarray(x)), O._p(np.ascontiguousarray(mp)), t, 4, O._p(mo))
    assert np.array_equal(mt, mo)
    c0 = np.cov(Xm[:5].T)
    Ik = np.eye(4) * 1e-4
    co = np.ascontiguousarray(c0.copy())
    O.lib().fmcmc_oracle_cov_recursive(O._p(np.ascontiguousarray(x)), O._p(co), O._p(np.ascontiguousarray(mp)), O._p(mo), t, 1e-5, 1.0,
                                       O._p(np.ascontiguousarray(Ik)), 4)
    assert np.allclose(f.cov_recursive(x, c0, mp, t, Mean_t=mt, eps=1e-5, Ik=Ik), co, rtol=1e-14, atol=1e-16)


def test_reflect_on_boundaries_matches_oracle(O):
    import ctypes as C
    rng = np.random.default_rng(3)
    lb, ub = np.array([-1.0, 0.0, 2.0, -5.0]), np.array([1.0, 0.5, 7.0, 5.0])
    which = np.array([0, 2, 3])
    for _ in range(200):
        x = rng.uniform(-40, 40, 4)
        got = f.reflect_on_boundaries(x, lb, ub, which)
        ref = x.copy()
        w32 = which.astype(np.int32)
        O.lib().fmcmc_oracle_reflect(O._p(ref), O._p(lb), O._p(ub), w32.ctypes.data_as(C.POINTER(C.c_int32)), 3, O.MATH_R)
        assert np.allclose(got, ref, atol=1e-12) and got[1] == x[1]
        assert np.all(got[which] >= lb[which]) and np.all(got[which] <= ub[which])


def test_plan_update_sequence_matches_the_reference_rules():
    """R/kernel.R:66-133: row r (1-based) of "ordered" / explicit plans updates seq[(r - 1) mod len]."""
    P = f.plan_update_sequence
    assert P(3, 4, False, "joint").all() and not P(3, 4, [False, True, False], "joint")[:, 1].any()
    o = P(3, 5, [False, True, False], "ordered")
    assert [list(np.nonzero(r)[0]) for r in o] == [[0], [2], [0], [2], [0]]
    e = P(3, 5, False, [2, 1, 3])
    assert [int(np.nonzero(r)[0][0]) for r in e] == [1, 0, 2, 1, 0]
    r = P(4, 200, [False, True, False, False], "random", rng=np.random.default_rng(1))
    assert np.all(r.sum(axis=1) == 1) and not r[:, 1].any() and r[:, [0, 2, 3]].any(axis=0).all()
    with pytest.raises(ValueError, match="cannot be zero"):
        P(2, 3, True, "joint")
    with pytest.raises(ValueError, match="same length"):
        P(3, 3, False, [1, 2])
    with pytest.raises(ValueError, match="-scheme- update must be"):
        P(3, 3, False, "sideways")


def _ar1(phi, n, seed):
    rng = np.random.default_rng(seed)
    e = rng.standard_normal(n)
    y = np.zeros(n)
    for t in range(1, n):
        y[t] = phi * y[t - 1] + e[t]
    return y


def test_spectrum0_ar_and_yule_walker():
    """coda::spectrum0.ar = var.pred / (1 - sum(ar))^2 of stats::ar(aic = TRUE); Levinson-Durbin == the Toeplitz solve."""
    from fmcmc_amd.convergence import _ar_yw, _pcramer
    y = _ar1(0.6, 40000, 1)
    s0, order = f.spectrum0_ar(y)
    assert order >= 1 and abs(s0 - 1 / 0.4 ** 2) / 6.25 < 0.12
    ar, vp, order = _ar_yw(y)
    n = y.size
    x = y - y.mean()
    r = np.array([np.dot(x[:n - l], x[l:]) / n for l in range(order + 1)])
    T = np.array([[r[abs(i - j)] for j in range(order)] for i in range(order)])
    assert np.allclose(ar, np.linalg.solve(T, r[1:]), atol=1e-10)
    assert np.isclose(vp, (r[0] - ar @ r[1:]) * n / (n - (order + 1)), rtol=1e-10)
    assert f.spectrum0_ar(np.full(50, 3.0)) == (0.0, 0) and f.spectrum0_ar(2.0 + 0.5 * np.arange(50.0)) == (0.0, 0)
    # Cramer-von Mises distribution: published 5 % and 1 % critical values 0.461 and 0.743
    assert abs(_pcramer(0.461) - 0.95) < 2e-3 and abs(_pcramer(0.743) - 0.99) < 2e-3


def test_geweke_and_heidel_checkers():
    """R/convergence.R:248-344 on host chains: stationary chains pass, drifting ones do not; error texts."""
    n = 4000
    it = np.arange(1, n + 1)
    stat = np.stack([5 + _ar1(0.5, n, 2), -3 + _ar1(0.2, n, 3)], axis=1)
    drift = stat + np.linspace(0, 6, n)[:, None]
    z = f.geweke_diag(stat, it)
    assert z.shape == (2,) and np.all(np.abs(z) < 3.5) and np.all(np.abs(f.geweke_diag(drift, it)) > 5)
    gw = f.convergence_geweke(200)
    assert gw(f.Mcmc(stat, start=1, end=n, thin=1)) is True and "avg Geweke's Z" in gw.msg and len(gw.history) == 1
    # the reference compares 1 - p-value with the threshold (R/convergence.R:283-288), so a drifting chain (|z| large) passes
    # -- kept as it is; a non-finite z (both windows constant) does not
    assert gw(f.Mcmc(drift, start=1, end=n, thin=1)) is True
    step = (np.arange(n) >= n // 3).astype(float)[:, None]
    assert not np.isfinite(f.geweke_diag(step, it)[0]) and gw(f.Mcmc(step, start=1, end=n, thin=1)) is False
    h = f.heidel_diag(stat, it)
    assert h.shape == (2, 6) and np.all(h[:, 0] == 1) and np.all(h[:, 3] == 1) and np.allclose(h[:, 4], stat.mean(0), atol=0.2)
    hd = f.convergence_heildel(500)
    assert hd(f.Mcmc(stat, start=1, end=n, thin=1)) is True and "Heidel's Avg. pval" in hd.msg
    assert hd(f.Mcmc(np.cumsum(stat, axis=0), start=1, end=n, thin=1)) is False          # random walk: not stationary
    two = f.McmcList([f.Mcmc(stat, start=1, end=n, thin=1)] * 2)
    with pytest.raises(ValueError, match="only available with runs of a single chain"):
        gw(two)
    with pytest.raises(ValueError, match="only available with runs of a single chain"):
        hd(two)
    with pytest.warns(UserWarning, match="failed to be computed"):
        assert gw(f.Mcmc(np.ones((1, 1)), start=1, end=1, thin=1)) is False
    thinned = f.geweke_diag(stat[::10], it[::10])      # window() works on iteration labels, not row numbers
    assert thinned.shape == (2,) and np.all(np.isfinite(thinned))
