"""include/fmh_detmath.h and include/fmh_philox.h checked INDEPENDENTLY of the engine:
accuracy against libm / scipy, and the published Philox4x32-10 known-answer vectors."""
import ctypes as C

import numpy as np
import pytest
from scipy.special import ndtri


def _eval(O, which, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    O.lib().fmcmc_oracle_detmath(which, O._p(x), O._p(out), x.size)
    return out


def _ulps(got, ref_ld):
    ref = ref_ld.astype(np.float64)
    return np.abs(got.astype(np.longdouble) - ref_ld) / np.spacing(np.abs(ref)).astype(np.longdouble)


@pytest.mark.parametrize("which,fn,gen", [
    (0, np.log, lambda r: np.exp(r.uniform(-700, 700, 200000))),
    (0, np.log, lambda r: r.uniform(0.5, 2.0, 200000)),
    (0, np.log, lambda r: r.uniform(1e-320, 1e-300, 20000)),          # subnormals
    (1, np.exp, lambda r: r.uniform(-745, 709, 200000)),
    (1, np.exp, lambda r: r.uniform(-1, 1, 200000)),
    (2, np.log1p, lambda r: np.concatenate([r.uniform(-0.999, 10, 100000), r.uniform(-1e-3, 1e-3, 100000),
                                            np.exp(r.uniform(-40, 0, 50000)), r.uniform(1, 1e6, 20000)])),
])
def test_detmath_accuracy(O, which, fn, gen):
    x = gen(np.random.default_rng(which))
    got = _eval(O, which, x)
    assert _ulps(got, fn(x.astype(np.longdouble))).max() < 1.0  # faithfully rounded


def test_detmath_special_values(O):
    inf, nan = np.inf, np.nan
    assert np.array_equal(_eval(O, 0, [0.0, 1.0, inf]), [-inf, 0.0, inf])
    assert np.isnan(_eval(O, 0, [-1.0, nan])).all()
    assert np.array_equal(_eval(O, 1, [-inf, 0.0, inf, 800.0, -800.0]), [0.0, 1.0, inf, inf, 0.0])
    assert np.array_equal(_eval(O, 2, [-1.0, 0.0, inf]), [-inf, 0.0, inf])
    assert np.isnan(_eval(O, 2, [-2.0])).all()
    assert _eval(O, 1, [-744.0])[0] > 0  # subnormal result survives the two-step scaling


def test_qnorm_matches_as241(O):
    rng = np.random.default_rng(3)
    p = np.concatenate([rng.uniform(0, 1, 200000), np.exp(rng.uniform(-36, 0, 50000)),
                        1 - np.exp(rng.uniform(-36, -1, 50000))])
    p = p[(p > 0) & (p < 1)]
    got = _eval(O, 3, p)
    ref = ndtri(p)
    assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)) < 4e-15
    # against the R-faithful evaluation order (plain Horner, libm): same algorithm, <= a few ulp apart
    rq = np.array([O.lib().r_qnorm_std(float(v)) for v in p[:20000]])
    assert np.max(np.abs(got[:20000] - rq) / np.abs(rq)) < 2e-15
    assert _eval(O, 3, [0.5])[0] == 0.0


def test_philox_known_answers(O):
    """Random123 kat_vectors for philox4x32-10."""
    out = (C.c_uint32 * 4)()
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, exp in kats:
        O.lib().fmcmc_oracle_philox(*ctr, *key, out)
        assert tuple(out) == exp


def test_canonical_draws_are_sane(O):
    L = O.lib()
    n = 20000
    lu = np.array([L.fmcmc_oracle_canon_draw(0, 7, i, 3, 0, 0.0) for i in range(n)])
    z = np.array([L.fmcmc_oracle_canon_draw(1, 7, i, 3, i % 5, 0.0) for i in range(n)])
    t5 = np.array([L.fmcmc_oracle_canon_draw(2, 7, i, 3, i % 5, 5.0) for i in range(n)])
    assert np.all(lu < 0) and abs(np.mean(-lu) - 1.0) < 0.03            # -log U ~ Exp(1)
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1) < 0.03
    assert abs(t5.mean()) < 0.05 and abs(t5.var() - 5 / 3) < 0.15       # var of t_5 = 5/3
    # counter-based: same index -> same value, different chain -> different value
    assert L.fmcmc_oracle_canon_draw(1, 7, 10, 3, 0, 0.0) == L.fmcmc_oracle_canon_draw(1, 7, 10, 3, 0, 0.0)
    assert L.fmcmc_oracle_canon_draw(1, 7, 10, 3, 0, 0.0) != L.fmcmc_oracle_canon_draw(1, 7, 10, 4, 0, 0.0)


def test_logit_g_accuracy(O):
    """fmh_logit_g(eta) = log(2 cosh(eta / 2)) = |eta| / 2 + log1p(exp(-|eta|)), the per-observation term of the logistic
    family (per-row degree-5 polynomials on the grid 1/64, include/fmh_detmath.h): within 1.05 ulp of a 60-digit reference (measured 1.011)
    everywhere (rounding of c0 + the last fma; the composition of two libm calls that R evaluates reaches 1.5), even,
    monotone in |eta| across the table's rows, |eta| / 2 beyond the table, NaN in NaN out -- and through it the canonical
    log-likelihood equals R's form sum(logp[y == 1]) + sum(logq[y == 0]) (vignettes/workflow-with-fmcmc.Rmd:35-41)."""
    from decimal import Decimal, getcontext
    getcontext().prec = 60
    rng = np.random.default_rng(5)
    L = O.lib()

    def g(x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        L.fmcmc_oracle_detmath(11, O._p(x), O._p(out), x.size)
        return out

    # (a) exact reference
    x = np.concatenate([np.exp(rng.uniform(-19, 6.5, 30000)), rng.uniform(0, 40, 20000), rng.uniform(0, 2, 10000),
                        np.arange(0, 2401) / 64.0, (np.arange(0, 2400) + 0.5) / 64.0,            # row starts and middles
                        np.nextafter(np.arange(1, 2401) / 64.0, 0.0),                             # the last argument of every row
                        [700.0, 3.7252902984619140625e-09, 1e-8, 0.5, 1.0, 36.7, 37.5, 37.49999999999999, 37.50000000000001, 50.0,
                         0.0, 5e-324, 1e-300, 1e300]])
    got = g(x)
    one, two = Decimal(1), Decimal(2)
    worst = 0.0
    for v, gv in zip(x, got):
        d = Decimal(float(v))
        e = (-d).exp() if v < 1000 else Decimal(0)
        r = d / two + (e * (one - e / 2 + e * e / 3) if v > 40 else (one + e).ln())
        worst = max(worst, abs(float((Decimal(float(gv)) - r) / Decimal(float(np.spacing(float(r)))))))
    assert worst <= 1.05, worst      # measured 1.011
    # (b) against libm, bulk; the function is even
    xb = np.concatenate([np.exp(rng.uniform(-45, 6.5, 2_000_000)), rng.uniform(0, 40, 500_000)])
    gb = g(xb)
    ref = 0.5 * xb + np.log1p(np.exp(-xb))
    assert np.max(np.abs(gb - ref) / np.spacing(ref)) <= 2.5
    assert np.array_equal(g(-xb[:100000]).view(np.uint64), gb[:100000].view(np.uint64))
    assert np.all(gb >= np.log(2.0))
    # (c) monotone on a fine grid across the rows and the end of the table
    xs = np.linspace(0.0, 45.0, 400_001)
    assert np.all(np.diff(g(xs)) >= 0)
    # (d) beyond the table g = |eta| / 2 exactly; NaN propagates; infinities stay infinite
    edge = np.array([37.5, 40.0, 1e10, 1e300, np.inf, -np.inf])     # (64 |eta| overflows beyond 2.8e306: inf, like eta itself a little later)
    assert np.array_equal(g(edge), 0.5 * np.abs(edge))
    assert np.isnan(g(np.array([np.nan]))[0])
    # (e) the canonical log-likelihood against R's form, both through the oracle
    n = 20000
    X = rng.standard_normal((n, 3))
    beta = np.array([-1.0, 0.5, -0.5, 0.25])
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-(beta[0] + X @ beta[1:])))).astype(np.float64)
    m = O.Model(O.FAM_LOGISTIC, X, y, intercept=True, guard=False, prior_div=8.0)
    for th in (beta, beta + 0.3, 8.0 * beta, np.zeros(4)):
        a, b = m.logpost(th, O.MATH_CANON), m.logpost(th, O.MATH_R)
        assert abs(a - b) <= 2e-12 * abs(b), (a, b)
