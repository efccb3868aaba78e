"""CPU tests of the SURVEY section 8(f) rank-3 rows: uniform kernels (R/kernel_unif.R), the update schemes of
plan_update_sequence (R/kernel.R:66-133) and kernel_ram's freq / constr (R/kernel_ram.R:129,149-150) -- oracle
restatement, host-side constructors and C-ABI validation.  The GPU parity of the same features is in
test_gpu_parity.py."""
import ctypes as C

import numpy as np
import pytest

from conftest import synth_linreg


@pytest.fixture(scope="module")
def abi():
    from fmcmc_amd import _abi, build
    if build.needs_build():
        build.build()
    _abi.lib()
    return _abi


def test_r_sample_known_answers(O):
    """sample.int(n, size, TRUE) under R >= 3.6 ("Rejection" sampling): outputs printed by R for these seeds."""
    assert list(O.RRng(123).sample_int(10, 5)) == [3, 3, 10, 2, 6]
    assert list(O.RRng(42).sample_int(10, 10)) == [1, 5, 1, 9, 10, 4, 2, 10, 1, 8]
    assert list(O.RRng(1).sample_int(5, 10)) == [1, 4, 1, 2, 5, 3, 2, 3, 3, 1]
    g = O.RRng(9)                                  # n = 1 still consumes one uniform per draw (rbits(0))
    assert list(g.sample_int(1, 4)) == [1, 1, 1, 1] and g.count == 4


def _iid(O, n=50, seed=3):
    rng = np.random.default_rng(seed)
    return O.Model(O.FAM_IID_NORMAL, None, rng.normal(1.0, 2.0, n))


def test_unif_kernel_replays_R_draw_order(O):
    """R/kernel_unif.R:70-76 inside R/mcmc.R:720-838: runif(nsteps) first, then k uniforms per step,
    theta1 = theta0 + (min. + (max. - min.) * u)."""
    m = _iid(O)
    kn = O.Kernel(O.K_UNIF, 2, min_=[-0.3, -0.2], max_=[0.4, 0.1])
    r = O.run(m, kn, initial=[1.0, 2.0], nsteps=60, rng_mode=O.RNG_RMT, math_mode=O.MATH_R, rng=O.RRng(5))
    g = O.RRng(5)
    logu = np.log(g.runif(60))
    th0 = np.array([1.0, 2.0]); f0 = m.logpost(th0, O.MATH_R)
    for i in range(2, 61):
        u = g.runif(2)
        th1 = th0 + (kn.mu + kn.scale * u)
        assert np.array_equal(r.draws[0, i - 1], th1)
        f1 = m.logpost(th1, O.MATH_R)
        if logu[i - 1] < f1 - f0:
            th0, f0 = th1, f1
        assert np.array_equal(r.samples[0, i - 1], th0)


def test_unif_kernels_reference_properties(O):
    """inst/tinytest/test-kernel_unif.R: increments stay in [min., max.], fixed parameters never move, reflective
    proposals stay inside [lb, ub], "ordered" alternates the free parameters."""
    X, y = synth_linreg(300, 2, 1)
    m = O.Model(O.FAM_LINREG, X, y)
    lb, ub = -1.0, 0.5
    kn = O.Kernel(O.K_UNIF, 4, min_=lb, max_=ub, fixed=[False, True, False, False])
    r = O.run(m, kn, initial=[0, 0, 0, 3.0], nsteps=3000, seed=3)
    inc = r.draws[0, 1:] - r.samples[0, :-1]
    assert np.all(inc[:, 1] == 0) and np.all(inc[:, [0, 2, 3]] >= lb) and np.all(inc[:, [0, 2, 3]] <= ub)
    assert abs(inc[:, [0, 2, 3]].mean() - (lb + ub) / 2) < 0.025
    kr = O.Kernel(O.K_UNIF_REFLECTIVE, 4, min_=lb, max_=ub, lb=[-0.5, -0.5, -0.5, 2.5], ub=[0.25, 0.25, 0.25, 3.5])
    r = O.run(m, kr, initial=[0, 0, 0, 3.0], nsteps=3000, seed=3)
    assert np.all(r.draws[0, 1:] >= kr.lb) and np.all(r.draws[0, 1:] <= kr.ub)
    ko = O.Kernel(O.K_UNIF, 4, min_=lb, max_=ub, fixed=[False, True, False, True], scheme="ordered")
    r = O.run(m, ko, initial=[0, 0, 0, 3.0], nsteps=400, seed=3)
    inc = r.draws[0, 1:] - r.samples[0, :-1]            # row i - 2 = loop step i: which(!fixed)[(i - 1) mod 2]
    steps = np.arange(2, 401)
    assert np.all(inc[(steps - 1) % 2 == 0][:, [1, 2, 3]] == 0) and np.all(inc[(steps - 1) % 2 == 1][:, [0, 1, 3]] == 0)
    with pytest.raises(ValueError, match="-max.- cannot be <= than -min.-."):
        O.Kernel(O.K_UNIF, 2, min_=1.0, max_=1.0)
    kd = O.Kernel(O.K_UNIF_REFLECTIVE, 2, min_=-2.0, max_=3.0)   # lb / ub default to min. / max.
    assert np.all(kd.lb == -2.0) and np.all(kd.ub == 3.0)


def test_random_scheme_is_R_sample_and_belongs_to_the_kernel(O):
    """R/kernel.R:106-113: sample(which(!fixed), nsteps, TRUE) drawn at the first proposal, reused by later calls."""
    m = _iid(O)
    kn = O.Kernel(O.K_NORMAL, 2, scale=0.3, scheme="random")
    st = O.ChainState(np.array([[1.0, 2.0]] * 2), 2)
    g = O.RRng(11)
    r1 = O.run(m, kn, nsteps=50, state=st, rng_mode=O.RNG_RMT, math_mode=O.MATH_R, rng=g)
    h = O.RRng(11)
    for c in range(2):
        h.runif(50)
        assert np.array_equal(st.scheme_cols[c], h.sample_int(2, 50) - 1)
        h.rnorm(49)
    assert h.count == g.count
    plan = st.scheme_cols.copy()
    n0 = g.count
    r2 = O.run(m, kn, nsteps=50, state=st, rng_mode=O.RNG_RMT, math_mode=O.MATH_R, rng=g)
    assert np.array_equal(st.scheme_cols, plan) and g.count - n0 == 2 * (50 + 2 * 49)     # no second sample(); inversion rnorm takes two uniforms
    for r in (r1, r2):
        inc = r.draws[:, 1:] != r.samples[:, :-1]
        assert np.array_equal(np.argmax(inc, axis=2), plan[:, 1:]) and np.all(inc.sum(axis=2) == 1)
    O.run(m, kn, nsteps=30, state=st, rng_mode=O.RNG_RMT, math_mode=O.MATH_R, rng=g)         # shorter last bulk
    with pytest.raises(IndexError, match="subscript out of bounds"):
        O.run(m, kn, nsteps=51, state=st, rng_mode=O.RNG_RMT, math_mode=O.MATH_R, rng=g)


def test_random_scheme_canonical_plan(O):
    """PHILOX mode: column = pool[(word * npool) >> 32] of counter (row i, global chain, 0, SCHEME); the same in every
    call and for every sharding; a single free parameter at position j samples 1:j like R does."""
    m = _iid(O)
    kn = O.Kernel(O.K_NORMAL_REFLECTIVE, 2, scale=0.3, lb=[-50, 0.1], ub=50, scheme="random")
    st = O.ChainState(np.array([[1.0, 2.0]] * 3), 2)
    O.run(m, kn, nsteps=40, seed=99, chain_base=5, state=st)
    want = np.array([[O.lib().fmcmc_oracle_canon_draw(4, 99, i, 5 + c, 2, 0.0) for i in range(1, 41)] for c in range(3)])
    assert np.array_equal(st.scheme_cols, want.astype(np.int32))
    plan = st.scheme_cols.copy()
    O.run(m, kn, nsteps=40, seed=99, chain_base=5, state=st)
    assert np.array_equal(st.scheme_cols, plan)
    st1 = O.ChainState(np.array([[1.0, 2.0]]), 2)
    O.run(m, kn, nsteps=40, seed=99, chain_base=7, state=st1)
    assert np.array_equal(st1.scheme_cols[0], plan[2])
    k1 = O.Kernel(O.K_NORMAL, 2, scale=0.3, fixed=[True, False], scheme="random")
    s1 = O.ChainState(np.array([[1.0, 2.0]]), 1)
    O.run(m, k1, nsteps=200, seed=1, state=s1)
    assert set(np.unique(s1.scheme_cols[0])) == {0, 1}            # sample(2, ...) == sample(1:2, ...)


def test_explicit_scheme(O):
    """R/kernel.R:69-92: scheme = c(2, 1, 3) recycled along the rows; validation texts."""
    X, y = synth_linreg(200, 1, 2)
    m = O.Model(O.FAM_LINREG, X, y)
    kn = O.Kernel(O.K_NORMAL, 3, scale=0.1, scheme=[2, 1, 3])
    r = O.run(m, kn, initial=[0, 0, 2.0], nsteps=100, seed=4)
    inc = r.draws[0, 1:] != r.samples[0, :-1]
    want = np.array([[2, 1, 3][(i - 1) % 3] - 1 for i in range(2, 101)])
    assert np.array_equal(np.argmax(inc, axis=1), want) and np.all(inc.sum(axis=1) == 1)
    with pytest.raises(ValueError, match="same length"):
        O.Kernel(O.K_NORMAL, 3, scheme=[1, 2])
    with pytest.raises(ValueError, match="not included in the ordering sequence"):
        O.Kernel(O.K_NORMAL, 3, scheme=[1, 1, 2])


def test_ram_freq_and_constr_oracle(O):
    X, y = synth_linreg(300, 2, 6)
    m = O.Model(O.FAM_LINREG, X, y)
    init = np.array([[0, 0, 0, 3.0]])
    base = O.run(m, O.Kernel(O.K_RAM, 4), initial=init, nsteps=120, seed=8)
    ones = O.run(m, O.Kernel(O.K_RAM, 4, constr=np.ones((4, 4))), initial=init, nsteps=120, seed=8)
    assert np.array_equal(base.samples, ones.samples) and np.array_equal(base.state.Sigma, ones.state.Sigma)
    never = O.run(m, O.Kernel(O.K_RAM, 4, freq=1000), initial=init, nsteps=120, seed=8)    # i %% freq never 0
    assert np.array_equal(never.state.Sigma[0], 1e-4 * np.eye(4)) and never.state.abs_iter[0] == 119
    diag = O.run(m, O.Kernel(O.K_RAM, 4, constr=np.eye(4)), initial=init, nsteps=120, seed=8)
    S = diag.state.Sigma[0]
    assert np.all(S[~np.eye(4, dtype=bool)] == 0) and np.all(np.diag(S) > 0)
    fx = O.run(m, O.Kernel(O.K_RAM, 4, fixed=[False, True, False, False], constr=np.eye(4)), initial=init, nsteps=60, seed=8)
    assert fx.state.Sigma.shape == (1, 3, 3)                  # constr[which., which.]
    for math, rng in ((O.MATH_R, O.RRng(2)),):                # R arithmetic: freq gates the same steps
        r = O.run(m, O.Kernel(O.K_RAM, 4, freq=1000), initial=init, nsteps=50, rng_mode=O.RNG_RMT, math_mode=math, rng=rng)
        assert np.array_equal(r.state.Sigma[0], 1e-4 * np.eye(4))


def test_ram_qfun_eta_families_oracle(O):
    """fmcmc_kernel.ram_qfun / ram_df / ram_eta_exp: zeros are the defaults (rt(k, k), exponent 2/3); each family changes
    the stream; in R arithmetic the variates come from norm_rand / rt of the restated generator."""
    X, y = synth_linreg(300, 2, 6)
    m = O.Model(O.FAM_LINREG, X, y)
    init = np.array([[0, 0, 0, 3.0]])
    base = O.run(m, O.Kernel(O.K_RAM, 4), initial=init, nsteps=150, seed=8)
    same = O.run(m, O.Kernel(O.K_RAM, 4, ram_qfun=2, ram_df=4.0, ram_eta_exp=2.0 / 3.0), initial=init, nsteps=150, seed=8)
    assert np.array_equal(base.samples, same.samples) and np.array_equal(base.state.Sigma, same.state.Sigma)
    seen = [base.samples.tobytes()]
    for kw in (dict(ram_qfun=1), dict(ram_qfun=2, ram_df=2.5), dict(ram_eta_exp=0.9)):
        r = O.run(m, O.Kernel(O.K_RAM, 4, **kw), initial=init, nsteps=150, seed=8)
        assert r.samples.tobytes() not in seen and np.all(np.isfinite(r.state.Sigma))
        seen.append(r.samples.tobytes())
    # qfun = rnorm in R arithmetic: the proposal of row 2 is theta0 + eps * norm_rand() draws taken after runif(nsteps)
    g = O.RRng(5)
    r = O.run(m, O.Kernel(O.K_RAM, 4, ram_qfun=1, warmup=1000), initial=init, nsteps=5, rng_mode=O.RNG_RMT, math_mode=O.MATH_R, rng=g)
    g2 = O.RRng(5); g2.runif(5)
    z = g2.rnorm(4)
    assert np.allclose(r.draws[0, 1], init[0] + 1e-4 * z, rtol=0, atol=1e-15)
    import fmcmc_amd as f
    from fmcmc_amd import _abi as abi
    k = f.kernel_ram(qfun=f.qfun_normal(), eta=f.eta_power(0.75))
    assert k.ram_qfun == abi.RAM_QFUN_NORMAL and k.ram_eta_exp == 0.75 and k.eta(8, 2) == min(1.0, 8 ** -0.75 * 2)
    k = f.kernel_ram(qfun=f.qfun_t(3))
    assert k.ram_qfun == abi.RAM_QFUN_T_DF and k.ram_df == 3.0
    assert f.kernel_ram(qfun=f.qfun_t()).ram_qfun == abi.RAM_QFUN_T_K
    with pytest.raises(NotImplementedError, match="closure"):
        f.kernel_ram(qfun=lambda k: np.zeros(k))
    with pytest.raises(NotImplementedError, match="closure"):
        f.kernel_ram(eta=lambda i, k: 0.5)
    with pytest.raises(ValueError):
        f.eta_power(0.0)


def test_host_kernel_constructors(monkeypatch):
    import fmcmc_amd as f
    from fmcmc_amd import _abi as abi
    k = f.kernel_unif(min_=[-1, -2, -3], max_=[1, 2, 4], fixed=[False, True, False], scheme="ordered")
    k._init(3)
    assert k.k == 1 and np.array_equal(k.mu, [-1, -2, -3]) and np.array_equal(k.scale, [2, 4, 7]) and k._scheme_id == abi.SCHEME_ORDERED
    k = f.kernel_unif_reflective(min_=-1.0, max_=0.5)
    k._init(2)
    assert np.all(k.lb == -1.0) and np.all(k.ub == 0.5) and k.k == 2
    with pytest.raises(ValueError, match="-max.- cannot be <= than -min.-."):
        f.kernel_unif(min_=1.0, max_=0.0)._init(2)
    with pytest.raises(ValueError, match="-ub- cannot be <= than -lb-."):
        f.kernel_unif_reflective(lb=1.0, ub=1.0)._init(2)
    k = f.kernel_normal(scheme=[2, 1, 3]); k._init(3)
    assert k._scheme_id == abi.SCHEME_EXPLICIT and list(k._scheme_seq) == [1, 0, 2] and k.k == 1
    k = f.kernel_normal(scheme="random"); k._init(3)
    assert k._scheme_id == abi.SCHEME_RANDOM and k.k == 1
    with pytest.raises(ValueError, match="same length"):
        f.kernel_normal(scheme=[1, 2])._init(3)
    with pytest.raises(ValueError, match="not included in the ordering sequence"):
        f.kernel_normal(scheme=[1, 3], fixed=[False, False, True])._init(3)
    with pytest.raises(ValueError, match="-scheme- update must be"):
        f.kernel_normal(scheme="zigzag")._init(3)
    assert f.kernel_ram(freq=3, constr=np.eye(2)).freq == 3
    with pytest.raises(ValueError, match="freq"):
        f.kernel_ram(freq=0)


def test_abi_validation_of_the_new_rows(abi):
    """fmcmc_validate (host pointers, no GPU): the reference's stop() texts for the new arguments."""
    L = abi.lib()
    y = np.zeros(10); X = np.zeros((1, 10))
    m = abi.Model(abi.FAM_GAUSSIAN_LINREG, 1, 10, X.ctypes.data, y.ctypes.data, 1, 1, 0.0)
    run = abi.Run(2, 100, 0, 1, 0, 0, 0, abi.RNG_PHILOX, 0, None, None)
    mu, lb, ub = np.zeros(3), np.full(3, -1e300), np.full(3, 1e300)
    fixed = np.zeros(3, dtype=np.uint8)

    def kern(kind, scale, scheme=0, seq=None, freq=1):
        sc = np.asarray(scale, dtype=np.float64)
        sq = None if seq is None else np.asarray(seq, dtype=np.int32)
        k = abi.Kernel(kind, 3, mu.ctypes.data, sc.ctypes.data, lb.ctypes.data, ub.ctypes.data, fixed.ctypes.data, scheme,
                       freq, 0, 0, float("inf"), 1e-4, 0.234, 0.0, sq.ctypes.data if sq is not None else None,
                       0 if sq is None else sq.size, 0, None)
        k._keep = (sc, sq)
        return k

    def check(k, code, text=None):
        rc = L.fmcmc_validate(C.byref(m), C.byref(k), C.byref(run))
        assert rc == code, abi.last_error()
        if text:
            assert text in abi.last_error()

    check(kern(abi.KERNEL_UNIF, [1, 1, 1]), abi.OK)
    check(kern(abi.KERNEL_UNIF, [1, 0, 1]), abi.ERR_ARG, "-max.- cannot be <= than -min.-.")
    check(kern(abi.KERNEL_UNIF_REFLECTIVE, [1, 1, 1], scheme=abi.SCHEME_RANDOM), abi.OK)
    check(kern(abi.KERNEL_NORMAL, [1, 1, 1], scheme=7), abi.ERR_ARG, "-scheme- update must be")
    check(kern(abi.KERNEL_NORMAL, [1, 1, 1], scheme=abi.SCHEME_EXPLICIT, seq=[2, 0, 1]), abi.OK)
    check(kern(abi.KERNEL_NORMAL, [1, 1, 1], scheme=abi.SCHEME_EXPLICIT, seq=[2, 0]), abi.ERR_ARG, "same length")
    check(kern(abi.KERNEL_NORMAL, [1, 1, 1], scheme=abi.SCHEME_EXPLICIT, seq=[2, 2, 1]), abi.ERR_ARG, "not included")
    check(kern(abi.KERNEL_RAM, [1, 1, 1], freq=5), abi.OK)
    check(kern(abi.KERNEL_RAM, [1, 1, 1], freq=0), abi.ERR_ARG, "freq")
    check(kern(abi.KERNEL_ADAPT, [1, 1, 1], freq=2), abi.OK)
    check(kern(abi.KERNEL_ADAPT, [1, 1, 1], freq=0), abi.ERR_ARG, "freq")
    check(kern(9, [1, 1, 1]), abi.ERR_ARG, "unknown kernel kind")


def test_adapt_window_and_stride_oracle(O):
    """kernel_adapt(bw > 0) and (freq > 1), R/kernel_adapt.R:117-160: the final Sigma recomputed offline from the chain with
    numpy / the package's own cov_recursive (matrix input = R's row-by-row recursion)."""
    import fmcmc_amd as f
    X, y = synth_linreg(400, 1, 12)
    m = O.Model(O.FAM_LINREG, X, y)
    init = np.array([[0.0, 0.0, 4.0]])
    for math, rng in ((O.MATH_CANON, None), (O.MATH_R, O.RRng(3))):
        kw = dict(rng_mode=O.RNG_RMT, rng=rng) if rng else {}
        # windowed: last step is a gated one (freq = 1), rows nsteps - bw + 1 .. nsteps - 1
        kn = O.Kernel(O.K_ADAPT, 3, bw=20, warmup=30, Sd=0.7)
        r = O.run(m, kn, initial=init, nsteps=120, seed=5, math_mode=math, **kw)
        rows = r.samples[0, 120 - 20:119]                          # 0-based rows of ans[(i-bw+1):(i-1)] for i = 120
        want = 0.7 * (np.cov(rows.T) + 1e-4 * np.eye(3))
        assert rows.shape[0] == 19 and np.allclose(r.state.Sigma[0], want, rtol=1e-9, atol=1e-15)
        # strided: every 4th step folds the 4 previous rows in, one by one
        kn = O.Kernel(O.K_ADAPT, 3, freq=4, warmup=10)
        r = O.run(m, kn, initial=init, nsteps=121, seed=5, math_mode=math, **kw)
        S, mean = 1e-4 * np.eye(3), None
        for i in range(2, 122):
            abs_iter = i - 2
            if abs_iter > 10 and i > 2 and i % 4 == 0:
                if mean is None:
                    mean = r.samples[0, :i - 1].mean(0)
                Xt = r.samples[0, i - 4 - 1:i - 1]                  # rows (i-freq):(i-1)
                means = f.mean_recursive(Xt, mean, abs_iter - 4)
                S = f.cov_recursive(Xt, S, mean, abs_iter - 4, Mean_t=means, eps=1e-5, Ik=1e-4 * np.eye(3))[-1]
                mean = means[-1]
        assert np.allclose(r.state.Sigma[0], S, rtol=1e-9, atol=1e-18) and np.allclose(r.state.mean_prev[0], mean, rtol=1e-12)
    # R fails when the window reaches before the first row of the call: continuation of a warmed-up windowed kernel
    kn = O.Kernel(O.K_ADAPT, 3, bw=20, warmup=30)
    st = O.ChainState(init, 3)
    O.run(m, kn, nsteps=60, seed=1, state=st)
    r = O.run(m, kn, nsteps=60, seed=1, state=st)
    assert r.status[0] == 4 and r.status_step[0] == 3
    with pytest.raises(ValueError, match="warmup"):
        f.kernel_adapt(bw=40, warmup=30)
    assert f.kernel_adapt(bw=10, freq=2).bw == 10


@pytest.mark.parametrize("kind_name", ["nmirror", "umirror"])
def test_mirror_kernels_replay_R_semantics(O, kind_name):
    """R/kernel_mirror.R inside R/mcmc.R:720-838, restated independently in numpy on R's stream: mu <- running mean while
    1 <= abs_iter <= warmup, scale rescaled once at abs_iter == nadapt (the closure's own argument), proposal
    N(2 mu - theta0, scale) resp. U(2 mu - theta0 -+ sqrt(3) scale)."""
    kind = O.K_NMIRROR if kind_name == "nmirror" else O.K_UMIRROR
    rng = np.random.default_rng(3)
    m = O.Model(O.FAM_IID_NORMAL, None, rng.normal(1.0, 2.0, 60))
    kn = O.Kernel(kind, 2, mu=[1.0, 2.0], scale=[0.4, 0.3], warmup=30, nadapt=4, lb=[-20, 0.05], ub=20.0)
    r = O.run(m, kn, initial=[1.0, 2.0], nsteps=80, rng_mode=O.RNG_RMT, math_mode=O.MATH_R, rng=O.RRng(6))
    g = O.RRng(6)
    logu = np.log(g.runif(80))
    th0 = np.array([1.0, 2.0]); f0 = m.logpost(th0, O.MATH_R)
    mu, scale, ans = np.array([1.0, 2.0]), np.array([0.4, 0.3]), [th0.copy()]
    for i in range(2, 81):
        a = i - 2
        if 1 <= a <= 30:
            mu = (mu * a + ans[-1]) / (a + 1)
        if a == 4:
            d = np.diff(np.array(ans), axis=0)
            obs = 1.0 - np.mean((d ** 2).sum(axis=1) == 0.0)
            scale = scale * np.tan(np.pi / 2.0 * obs) / np.tan(np.pi / 2.0 * 0.4)
        elif 4 < a <= 30:     # obs_arate <<- mean_recursive(as.double(ans[i-1, ] != ans[i-2, ]), obs_arate, abs_iter): a k-vector from here on
            obs = (obs * a + (ans[-1] != ans[-2]).astype(float)) / (a + 1)
        if kind == O.K_NMIRROR:
            th1 = (2 * mu - th0) + scale * g.rnorm(2)
        else:
            lo, hi = 2 * mu - th0 - np.sqrt(3.0) * scale, 2 * mu - th0 + np.sqrt(3.0) * scale
            th1 = lo + (hi - lo) * g.runif(2)
        ref = th1.copy()
        w = np.array([0, 1], dtype=np.int32)
        O.lib().fmcmc_oracle_reflect(O._p(ref), O._p(kn.lb), O._p(kn.ub), w.ctypes.data_as(C.POINTER(C.c_int32)), 2, O.MATH_R)
        assert np.allclose(r.draws[0, i - 1], ref, rtol=1e-13, atol=1e-15), i
        f1 = m.logpost(r.draws[0, i - 1], O.MATH_R)
        if logu[i - 1] < f1 - f0:
            th0, f0 = r.draws[0, i - 1].copy(), f1
        ans.append(th0.copy())
        assert np.array_equal(r.samples[0, i - 1], th0)
    assert np.allclose(r.state.mirror_mu[0], mu, rtol=1e-13) and np.allclose(r.state.mirror_scale[0], scale, rtol=1e-13)
    assert np.shape(obs) == (2,) and np.allclose(r.state.obs_arate[0], obs, rtol=0, atol=1e-15) and r.state.abs_iter[0] == 79


def test_mirror_kernels_sample_the_posterior(O):
    """inst/tinytest/test-kernel_mirror.R checks posterior means; same here on the README-style regression."""
    X, y = synth_linreg(500, 1, 4, beta=[3.0, 2.0])
    m = O.Model(O.FAM_LINREG, X, y)
    for kind in (O.K_NMIRROR, O.K_UMIRROR):
        kn = O.Kernel(kind, 3, mu=[2.5, 1.5, 3.5], scale=0.3, warmup=1000, nadapt=5, lb=[-50, -50, 0.01], ub=50.0)
        r = O.run(m, kn, initial=[[2.5, 1.5, 3.5]] * 2, nsteps=4000, seed=11)
        post = r.samples[:, 1000:].mean(axis=(0, 1))
        assert np.all(np.abs(post - [3.0, 2.0, 4.0]) < 0.5) and 0.05 < r.accept_count.mean() / 3999 < 0.95
