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


def test_fused_softplus_accuracy(O):
    """fmh_log1p_exp_nonpos, the softplus tail log1p(exp(x)), x <= 0, of the logistic family (grid form: softplus(a_j) +
    log1p(sigma_j expm1(r)), include/fmh_detmath.h): within 0.6 ulp of a 60-digit reference on its fast range [-37.5, 0]
    (measured 0.51, i.e. correctly rounded but for a few percent of the arguments; the composition of two faithfully
    rounded libm calls that R evaluates reaches 1.5), within 1.75 ulp everywhere, within 3 ulp of libm's composition on
    2.5M points, monotone where it should be, and equal to the general functions outside the fast range."""
    from decimal import Decimal, getcontext
    getcontext().prec = 60
    rng = np.random.default_rng(5)
    L = O.lib()

    def fused(x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        L.fmcmc_oracle_detmath(11, O._p(x), O._p(out), x.size)
        return out

    def composed(x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        L.fmcmc_oracle_detmath(9, O._p(x), O._p(out), x.size)
        return out

    # (a) exact reference
    x = np.concatenate([-np.exp(rng.uniform(-19, 6.5, 30000)), -rng.uniform(0, 40, 20000), -rng.uniform(0, 2, 10000),
                        -np.arange(0, 2401) / 64.0, -(np.arange(0, 2400) + 0.5) / 64.0,       # grid points and interval ends
                        [-700.0, -3.7252902984619140625e-09, -1e-8, -0.5, -1.0, -36.7, -37.5, -37.50000000000001, -50.0, -699.9,
                         -0.0, -5e-324, -1e-300]])
    x = x[x >= -700]
    got = fused(x)
    one = Decimal(1)
    worst, worst_fast = 0.0, 0.0
    for v, g in zip(x, got):
        e = Decimal(float(v)).exp()
        r = e * (one - e / 2 + e * e / 3) if v < -40 else (one + e).ln()
        err = abs(float((Decimal(float(g)) - r) / Decimal(float(np.spacing(float(r))))))
        worst = max(worst, err)
        if v >= -37.5:
            worst_fast = max(worst_fast, err)
    assert worst <= 1.75 and worst_fast <= 0.6, (worst, worst_fast)
    # (b) against libm's composition, bulk
    xb = np.concatenate([-np.exp(rng.uniform(-45, 6.5, 2_000_000)), -rng.uniform(0, 40, 500_000)])
    xb = xb[xb >= -700]
    gb = fused(xb)
    ref = np.log1p(np.exp(xb))
    assert np.max(np.abs(gb - ref) / np.spacing(ref)) <= 3.0
    assert np.all(gb > 0) and np.all(gb <= np.log(2.0))
    # (c) monotone on a fine grid across table boundaries and exponent changes of exp(x)
    xs = -np.linspace(1e-6, 45.0, 400_001)
    gs = fused(xs)
    assert np.all(np.diff(gs) <= 0)
    # (d) below -37.5, for positive arguments and NaN the general functions take over, bit for bit
    edge = np.array([-745.2, -746.0, -745.13321910194110842, -700.0000000000001, -709.0, -37.500000000000007, -40.0,
                     -1000 * np.log(2), np.nan, 2.0, 5e-324])
    assert np.array_equal(fused(edge).view(np.uint64), composed(edge).view(np.uint64))
