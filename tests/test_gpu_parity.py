"""GPU parity: the HIP engine (through the C-ABI) must be BIT-IDENTICAL to the oracle's
PHILOX/canonical mode on the same seeded inputs: samples, proposals, log-posteriors, accept
bitmap and the persistent kernel state.  All tests here need a real MI355X."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import synth_linreg, set_knob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from fmcmc_amd import engine, _abi
    _abi.lib()
    return engine


@pytest.fixture(autouse=True)
def _four_chains_per_workgroup(request, monkeypatch):
    """The cases of this module reach the kernels of FULL launches (four chains per workgroup: mh_sweep_mfma, mh_sweep_spec) with
    a handful of chains, so they switch the latency form off (knob lat=0).  It has its own tests (test_latency_form_*), and the
    randomised sweeps and everything in test_gpu_api.py run on the dispatcher's own choice."""
    if "latency" not in request.node.name and "randomised" not in request.node.name:
        set_knob(monkeypatch, "lat", "0")


def _bits_equal(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
    b = np.ascontiguousarray(b, dtype=np.float64).view(np.uint64)
    return np.array_equal(a, b)


class _Blocks:
    """The oracle on `threads` host threads: consecutive blocks of chains, each an oracle call with its chain_base (ctypes
    releases the GIL; chains never interact, R/mcmc.R:590-673), presented as ONE result / ONE state."""

    def __init__(self, O, om, ok, initial, threads):
        C_ = initial.shape[0]
        nb = max(1, min(threads, C_))
        edges = [C_ * b // nb for b in range(nb + 1)]
        self.O, self.om, self.ok = O, om, ok
        self.blocks = [(lo, hi, O.ChainState(initial[lo:hi], ok.kf)) for lo, hi in zip(edges[:-1], edges[1:]) if hi > lo]

    def run(self, **kw):
        from concurrent.futures import ThreadPoolExecutor
        base = kw.pop("chain_base", 0)
        with ThreadPoolExecutor(len(self.blocks)) as ex:
            rs = list(ex.map(lambda b: self.O.run(self.om, self.ok, chain_base=base + b[0], state=b[2], **kw), self.blocks))
        ro = type("R", (), {})()
        for name in ("status", "accept_bits", "accept_count", "samples_cks", "draws_cks", "logpost"):
            setattr(ro, name, np.concatenate([getattr(r, name) for r in rs], axis=0))
        return ro

    def __getattr__(self, name):     # the state arrays of all blocks
        vals = [getattr(b[2], name) for b in self.blocks]
        return None if vals[0] is None else np.concatenate(vals, axis=0)


def run_both(E, O, fam, X, y, kind, k, initial, nsteps, burnin=0, thin=1, seed=1215, chain_base=0,
             calls=1, intercept=True, guard=True, prior_div=0.0, threads=1, **kw):
    """Runs `calls` consecutive sweeps on GPU and oracle; asserts bit-equality after each.  threads > 1: the oracle runs
    blocks of chains on that many host threads (full-width configs)."""
    from fmcmc_amd import _abi as abi
    om = O.Model(fam, X, y, intercept=intercept, guard=guard, prior_div=prior_div)
    ok = O.Kernel(kind, k, **kw)
    gm = E.DeviceModel(fam, X, y, intercept=intercept, guard=guard, prior_div=prior_div)
    gk = E.KernelSpec(kind, k, ok.mu, ok.scale, ok.lb, ok.ub, ok.fixed, scheme=ok.scheme, freq=ok.freq,
                      warmup=ok.warmup, bw=ok.bw, until=ok.until, eps=ok.eps, arate=ok.arate, Sd=ok.Sd,
                      scheme_seq=ok.scheme_seq, constr=ok.constr, nadapt=ok.nadapt, ram_qfun=ok.ram_qfun, ram_df=ok.ram_df,
                      ram_eta_exp=ok.ram_eta_exp)
    initial = np.ascontiguousarray(initial, dtype=np.float64)
    blocks = _Blocks(O, om, ok, initial, threads) if threads > 1 else None
    ost = blocks if blocks is not None else O.ChainState(initial, ok.kf)
    gst = E.ChainState(initial, ok.kf)
    res = None
    for _ in range(calls):
        if blocks is not None:
            ro = blocks.run(nsteps=nsteps, burnin=burnin, thin=thin, seed=seed, chain_base=chain_base)
        else:
            ro = O.run(om, ok, nsteps=nsteps, burnin=burnin, thin=thin, seed=seed, chain_base=chain_base,
                       state=ost)
        rg = E.sweep(gm, gk, gst, nsteps, burnin=burnin, thin=thin, seed=seed, chain_base=chain_base,
                     check=False)
        import torch
        torch.cuda.synchronize()
        assert np.array_equal(rg.status.cpu().numpy(), ro.status)
        assert np.array_equal(rg.accept_bits.cpu().numpy().view(np.uint32), ro.accept_bits), "accept bitmap"
        assert np.array_equal(rg.accept_count.cpu().numpy(), ro.accept_count)
        good = ro.status == 0
        assert _bits_equal(rg.samples.cpu().numpy()[good], ro.samples_cks[good]), "samples"
        assert _bits_equal(rg.draws.cpu().numpy()[good], ro.draws_cks[good]), "draws"
        assert _bits_equal(rg.logpost.cpu().numpy()[good], ro.logpost[good]), "logpost"
        assert _bits_equal(gst.theta0.cpu().numpy(), ost.theta0)
        assert _bits_equal(gst.f0.cpu().numpy()[good], ost.f0[good])
        if kind in (O.K_ADAPT, O.K_RAM):
            assert np.array_equal(gst.abs_iter.cpu().numpy(), ost.abs_iter)
            assert _bits_equal(gst.Sigma.cpu().numpy(), ost.Sigma), "Sigma"
            assert np.array_equal(gst.nerrors.cpu().numpy(), ost.nerrors)
        if kind in (O.K_NMIRROR, O.K_UMIRROR):
            assert np.array_equal(gst.abs_iter.cpu().numpy(), ost.abs_iter)
            assert _bits_equal(gst.mirror_mu.cpu().numpy(), ost.mirror_mu), "mirror mu"
            assert _bits_equal(gst.mirror_scale.cpu().numpy(), ost.mirror_scale), "mirror scale"
            ga, oa = gst.obs_arate.cpu().numpy(), ost.obs_arate          # (NaN = R's NULL / numeric(0): payloads are not part of the spec)
            assert np.array_equal(np.isnan(ga), np.isnan(oa)) and _bits_equal(np.nan_to_num(ga, nan=-1.0), np.nan_to_num(oa, nan=-1.0)), "obs_arate"
        if ok.scheme == O.SCHEME_RANDOM:
            assert np.array_equal(gst.scheme_cols.cpu().numpy()[:, 1:nsteps], ost.scheme_cols[:, 1:nsteps]), "update plan"
        if kind == O.K_ADAPT:
            assert np.array_equal(gst.have_mean.cpu().numpy(), ost.have_mean)
            hm = ost.have_mean.astype(bool)
            assert _bits_equal(gst.mean_prev.cpu().numpy()[hm], ost.mean_prev[hm])
        res = (rg, ro)
    return res


def jitter_init(base, C, seed):
    rng = np.random.default_rng(seed)
    return np.asarray(base)[None, :] + 0.1 * rng.standard_normal((C, len(base)))


def test_detmath_device_equals_host(E, O):
    """include/fmh_detmath.h + fmh_philox.h: device bits == host bits, 1M points each."""
    import torch
    from fmcmc_amd import _abi as abi
    rng = np.random.default_rng(7)
    n = 1 << 20
    cases = {0: np.exp(rng.uniform(-700, 700, n)), 1: rng.uniform(-745, 709, n),
             2: np.concatenate([rng.uniform(-0.999, 50, n // 2), rng.uniform(-1e-3, 1e-3, n // 2)]),
             3: np.concatenate([rng.uniform(0, 1, n // 2), np.exp(rng.uniform(-36, 0, n // 2))]),
             4: np.zeros(n), 5: np.zeros(n), 6: rng.integers(1, 60, n).astype(np.float64),
             7: np.exp(rng.uniform(-700, 700, n)), 8: np.exp(rng.uniform(-700, 700, n)),
             # fused log1p(exp(x)), x <= 0: bulk, the tiny-|x| and underflow edges, and a few out-of-contract points
             9: np.concatenate([-np.exp(rng.uniform(-45, 6.7, n - 10)), [-0.0, 0.0, -745.2, -746.0, -3.7252902984619140625e-09,
                                                                         -745.13321910194110842, -700.0, -700.0000000000001, np.nan, 1.5]]),
             10: np.zeros(n)}
    for which, x in cases.items():
        x = np.ascontiguousarray(x)
        xd = torch.as_tensor(x).cuda()
        od = torch.empty_like(xd)
        rc = abi.lib().fmcmc_detmath_dev(which, xd.data_ptr(), od.data_ptr(), n, 99, None)
        assert rc == 0
        torch.cuda.synchronize()
        host = np.empty(n)
        if which < 4 or which == 9:   # (the host build of the fused softplus is function 11 of the oracle's table)
            O.lib().fmcmc_oracle_detmath(11 if which == 9 else which, O._p(x), O._p(host), n)
        else:
            O.lib().fmcmc_oracle_detmath_rng(which, O._p(x), O._p(host), n, 99)
        assert _bits_equal(od.cpu().numpy(), host), "detmath function %d differs between host and device" % which


@pytest.mark.parametrize("C,n,p", [(1, 200, 1), (3, 1000, 1), (5, 777, 3), (64, 2048, 3), (300, 1000, 2)])
def test_normal_linreg(E, O, C, n, p):
    X, y = synth_linreg(n, p, 11 + n)
    init = jitter_init([0.0] * (p + 1) + [float(np.std(y))], C, 5)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, p + 2, init, nsteps=300, scale=0.05)


def test_readme_first_run_philox(E, O, readme_data):
    X, y = readme_data
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 3, [[0, 0, O.r_sd(y)]] * 2, nsteps=2000)
    assert ro.accept_count.sum() > 0


def test_burnin_thin_and_continuation(E, O):
    X, y = synth_linreg(1500, 3, 3)
    init = jitter_init([0, 0, 0, 0, 4.0], 7, 1)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, init, nsteps=257, burnin=50, thin=7, calls=3,
             scale=0.03)


def test_chain_base_sharding_invariance(E, O):
    """chains [4,8) of an 8-chain job run as their own shard give the same bits."""
    X, y = synth_linreg(900, 2, 9)
    init = jitter_init([0, 0, 0, 4.0], 8, 2)
    rg_all, _ = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 4, init, nsteps=200, scale=0.05)
    rg_hi, _ = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 4, init[4:], nsteps=200, scale=0.05,
                        chain_base=4)
    assert _bits_equal(rg_all.samples.cpu().numpy()[4:], rg_hi.samples.cpu().numpy())


def test_fixed_and_ordered(E, O):
    X, y = synth_linreg(600, 2, 4)
    init = jitter_init([0, 0, 0, 4.0], 4, 3)
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 4, init, nsteps=300, scale=0.05,
                      fixed=[False, True, False, False])
    assert np.all(ro.samples[:, :, 1] == init[:, 1][:, None])  # fixed parameter never moves
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 4, init, nsteps=300, scale=0.1, scheme="ordered")


def test_reflective(E, O, readme_data):
    X, y = readme_data
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL_REFLECTIVE, 3, [[0, 0, O.r_sd(y)]] * 3,
                      nsteps=1500, guard=False, scale=0.5, ub=5.0, lb=[-5.0, 0.0, 0.0])
    d = ro.draws
    assert d[:, :, 0].min() >= -5 and d.max() <= 5 and d[:, :, 1:].min() >= 0


def test_adapt(E, O):
    X, y = synth_linreg(800, 3, 21)
    init = jitter_init([3, 2, -1, .5, 4.0], 6, 4)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, 5, init, nsteps=400, calls=2, warmup=100)


def test_ram(E, O):
    X, y = synth_linreg(800, 3, 22)
    init = jitter_init([3, 2, -1, .5, 4.0], 6, 5)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, 5, init, nsteps=400, calls=2)


def test_ram_bounded(E, O, readme_data):
    X, y = readme_data
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, 3, [[3, 2, 4.0]] * 4, nsteps=600, ub=[3.3, 5, 5],
             lb=[-5.0, 0.0, 0.0])


def test_ram_k20(E, O):
    X, y = synth_linreg(1000, 7, 23)
    init = jitter_init([3, 2, -1, .5, .25, -.75, 1.5, -2, 4.0], 5, 6)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, 9, init, nsteps=300)


def test_logistic(E, O):
    rng = np.random.default_rng(31)
    n, p = 1200, 4
    X = rng.standard_normal((n, p))
    beta = np.array([-1, .5, -.5, .25, -.25])
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-(beta[0] + X @ beta[1:])))).astype(np.float64)
    init = jitter_init(beta, 5, 8)
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL_REFLECTIVE, 5, init, nsteps=300, prior_div=8.0,
             scale=0.05, lb=-5.0, ub=5.0)


def test_iid_normal(E, O):
    rng = np.random.default_rng(1231)
    D = rng.normal(2.6, 3, 1000)
    run_both(E, O, O.FAM_IID_NORMAL, None, D, O.K_NORMAL_REFLECTIVE, 2, [[1.0, 1.0]] * 3, nsteps=500,
             guard=False, scale=0.1, lb=0.0, ub=10.0)


def test_iid_normal_on_the_linear_model_kernels(E, O, monkeypatch):
    """FAM_IID_NORMAL is the Gaussian linear model with an intercept and no covariate, in the oracle and in every kernel: since
    round 4 it takes that model's fast paths (normal / uniform kernels on the MFMA kernels, the adaptive ones on the streamed MFMA
    evaluation from 513 observations on) instead of the all-family kernel; since round 5 the adaptive and the mirror kernels run on the
    wave-specialised kernel up to 10,240 observations (compute lanes without an x; from one observation on)."""
    from fmcmc_amd import _abi as abi
    rng = np.random.default_rng(12)
    for n, want_n, want_a in ((60, "mfma", "spec"), (700, "mfma", "spec"), (9000, "mfma", "spec"), (20001, "mfma-streamed", "mfma-adaptive")):
        y = 1.5 + 2.0 * rng.standard_normal(n)
        init = jitter_init([1.0, 2.0], 7, 5)
        init[:, -1] = np.abs(init[:, -1]) + 0.1
        run_both(E, O, O.FAM_IID_NORMAL, None, y, O.K_NORMAL, 2, init, nsteps=90, burnin=4, thin=3, calls=2, scale=0.03)
        assert abi.last_kernel() == want_n
        run_both(E, O, O.FAM_IID_NORMAL, None, y, O.K_UNIF_REFLECTIVE, 2, init, nsteps=60, min_=-0.04, max_=0.05, lb=[-40.0, 0.05], ub=40.0)
        assert abi.last_kernel() == want_n
        run_both(E, O, O.FAM_IID_NORMAL, None, y, O.K_ADAPT, 2, init, nsteps=80, calls=2, warmup=10)
        assert abi.last_kernel() == want_a
        run_both(E, O, O.FAM_IID_NORMAL, None, y, O.K_RAM, 2, init, nsteps=80, calls=2, lb=[-40.0, 0.05], ub=40.0)
        assert abi.last_kernel() == want_a
        run_both(E, O, O.FAM_IID_NORMAL, None, y, O.K_RAM, 2, init, nsteps=80, burnin=2, thin=3)
        assert abi.last_kernel() == want_a
        run_both(E, O, O.FAM_IID_NORMAL, None, y, O.K_NMIRROR, 2, init, nsteps=120, calls=2, mu=[1.0, 2.0], scale=0.15, warmup=90, nadapt=6, lb=[-30.0, 0.05], ub=30.0)
        assert abi.last_kernel() == want_a
        if want_a == "spec" and n > 512:     # (knob specp0=0: the round-4 route)
            set_knob(monkeypatch, "specp0", "0")
            run_both(E, O, O.FAM_IID_NORMAL, None, y, O.K_ADAPT, 2, init, nsteps=80, calls=2, warmup=10)
            assert abi.last_kernel() == "mfma-adaptive"
            set_knob(monkeypatch, "specp0", "1")


def test_nan_logpost_is_reported(E, O):
    """README.md:356-361 ll without the guard: sigma < 0 -> NaN -> 'fun(par) is undefined'."""
    X, y = synth_linreg(300, 1, 2)
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 3, [[0, 0, 0.05]] * 2, nsteps=200,
                      guard=False, scale=1.0)
    assert (ro.status == 1).any()
    with pytest.raises(RuntimeError, match="undefined"):
        E.raise_on_chain_error(rg)


# ---- the register-resident variant of the headline shape (n = 10,000, 3 covariates, k = 5)
@pytest.mark.parametrize("kind", ["normal", "reflective", "adapt", "ram"])
def test_resident_headline_shape(E, O, kind):
    X, y = synth_linreg(10000, 3, 20260102)
    C = 9  # not a multiple of the 4 chains per workgroup
    init = jitter_init([0, 0, 0, 0, float(np.std(y))], C, 12)
    init[:, -1] = np.abs(init[:, -1])
    if kind == "normal":
        run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, init, nsteps=150, calls=2, scale=0.02)
    elif kind == "reflective":
        run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL_REFLECTIVE, 5, init, nsteps=150, scale=0.3,
                 lb=[-5, -5, -5, -5, 0.1], ub=[5, 5, 5, 5, 5.0], guard=False)
    elif kind == "adapt":
        run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, 5, init, nsteps=160, calls=2, warmup=40)
    else:
        run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, 5, init, nsteps=150, calls=2)


@pytest.mark.parametrize("n", [8193, 9000, 10240])
def test_resident_edge_sizes(E, O, n):
    """n range of the (P=3, OPT=20) variant: masks on the trailing observation slots."""
    X, y = synth_linreg(n, 3, n)
    init = jitter_init([0, 0, 0, 0, float(np.std(y))], 5, 13)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, init, nsteps=60, scale=0.02)


def test_streamed_equals_resident(E, O, monkeypatch):
    """Same bits from the streamed and the register-resident evaluation."""
    X, y = synth_linreg(10000, 3, 20260102)
    init = jitter_init([0, 0, 0, 0, float(np.std(y))], 6, 14)
    a, _ = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, init, nsteps=100, scale=0.02)
    set_knob(monkeypatch, "streamed", "1")
    b, _ = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, init, nsteps=100, scale=0.02)
    assert _bits_equal(a.samples.cpu().numpy(), b.samples.cpu().numpy())


# ---- the fp64 MFMA evaluation (default for 3 covariates, n in (9728, 10240], normal kernels)
@pytest.mark.parametrize("n", [9729, 10000, 10239, 10240])
def test_mfma_path_edge_sizes(E, O, n):
    X, y = synth_linreg(n, 3, 7 * n)
    init = jitter_init([0, 0, 0, 0, float(np.std(y))], 7, 21)
    init[:, -1] = np.abs(init[:, -1])
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, init, nsteps=90, burnin=7, thin=3, calls=2, scale=0.02)


def test_mfma_path_no_intercept_fixed_and_reflective(E, O):
    X, y = synth_linreg(10000, 3, 99)
    init = jitter_init([2, -1, .5, 5.0], 5, 22)          # theta = (b1, b2, b3, sigma), no intercept
    init[:, -1] = np.abs(init[:, -1])
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 4, init, nsteps=120, scale=0.02, intercept=False,
             fixed=[False, True, False, False])
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL_REFLECTIVE, 4, init, nsteps=120, scale=0.5, intercept=False,
             lb=[-3, -3, -3, 0.5], ub=[3, 3, 3, 6.0], guard=False)


def test_mfma_equals_valu_kernels(E, O, monkeypatch):
    """MFMA, wave-specialised VALU and streamed kernels: same bits (and == oracle inside run_both)."""
    X, y = synth_linreg(10000, 3, 20260102)
    init = jitter_init([0, 0, 0, 0, float(np.std(y))], 6, 23)
    a, _ = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, init, nsteps=80, scale=0.02)
    set_knob(monkeypatch, "mfma", "0")
    b, _ = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, init, nsteps=80, scale=0.02)
    assert _bits_equal(a.samples.cpu().numpy(), b.samples.cpu().numpy())
    from fmcmc_amd import _abi as abi
    assert abi.last_kernel().startswith("spec")


@pytest.mark.parametrize("n", [10000, 9000, 600])
@pytest.mark.parametrize("lat", ["0", "-1"])
def test_mfma_kernel_rare_paths(E, O, monkeypatch, n, lat):
    """The MFMA kernel (four chains per workgroup, knob lat=0) and what the dispatcher takes for 7 chains by itself (the
    latency form of the wave-specialised kernel): thinning, continuation, a fixed parameter, the uniform kernel and a chain
    that fails with a NaN (the rare-path branch) -- the oracle's bits in both."""
    from fmcmc_amd import _abi as abi
    set_knob(monkeypatch, "lat", lat)          # (-1: the dispatcher's own choice)
    X, y = synth_linreg(n, 3, 31 + n)
    init = jitter_init([0, 0, 0, 0, float(np.std(y))], 7, 24)
    init[:, -1] = np.abs(init[:, -1])
    bad = init.copy()
    bad[:, -1] = 0.03                                     # sigma steps below zero within a few proposals of scale 1
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, init, nsteps=150, burnin=11, thin=4, calls=2, scale=0.02)
    assert abi.last_kernel() == ("mfma" if lat == "0" else "lat1")
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, init, nsteps=90, scale=0.03, fixed=[False, False, True, False, False])
    run_both(E, O, O.FAM_LINREG, X, y, O.K_UNIF, 5, init, nsteps=90, min_=-0.03, max_=0.04)
    _, ro = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, bad, nsteps=60, guard=False, scale=1.0)
    assert (ro.status == 1).any()
    assert abi.last_kernel() == ("mfma" if lat == "0" else "lat1")


# ---- the LATENCY form (round 5): one, two or three chains per workgroup of the wave-specialised kernel
@pytest.mark.parametrize("C", [3, 256, 257, 512, 513, 768, 769])
@pytest.mark.parametrize("kind", ["normal", "reflective", "adapt", "ram"])
def test_latency_form_equals_the_oracle(E, O, kind, C):
    """Fewer than four chains per compute unit (R/mcmc.R:536-641 scales a FIXED number of chains over its workers, so a GPU
    of a sharded call holds nchains / G): the dispatcher gives each workgroup ceil(C / 256) chains -- 1 up to 256 chains, 2 up
    to 512, 3 up to 768, then the usual four.  C2 / C3's shape (n = 10,000, k = 5), every proposal family, the oracle's bits;
    the chain counts sit on both sides of every switch (and leave a last workgroup partly filled)."""
    from fmcmc_amd import _abi as abi
    X, y = synth_linreg(10000, 3, 20260102)
    init = jitter_init([0, 0, 0, 0, float(np.std(y))], C, 12)
    init[:, -1] = np.abs(init[:, -1])
    thr = 16
    if kind == "normal":
        run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, init, nsteps=40, calls=2, scale=0.02, threads=thr)
    elif kind == "reflective":
        run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL_REFLECTIVE, 5, init, nsteps=40, burnin=3, thin=2, scale=0.3,
                 lb=[-5, -5, -5, -5, 0.1], ub=[5, 5, 5, 5, 5.0], guard=False, threads=thr)
    elif kind == "adapt":
        run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, 5, init, nsteps=60, calls=2, warmup=20, threads=thr)
    else:
        run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, 5, init, nsteps=40, calls=2, threads=thr)
    per_cu = (C + 255) // 256
    simple = kind in ("normal", "reflective")          # (the normal kernels take the latency form up to two chains per workgroup)
    assert abi.last_kernel() == ((("lat%d" if simple else "spec-lat%d") % per_cu) if per_cu <= (2 if simple else 3) else ("mfma" if simple else "spec"))


@pytest.mark.parametrize("lat", ["1", "2", "3"])
@pytest.mark.parametrize("n,p", [(700, 1), (2000, 2), (5000, 4), (5120, 5), (3000, 6), (4096, 7), (10240, 3), (513, 3)])
def test_latency_form_shapes(E, O, monkeypatch, lat, n, p):
    """Every compute loop of the latency form (p = 1 .. 7 covariates, 2 .. 20 observation slots per lane, ragged last slots),
    forced to 1 / 2 / 3 chains per workgroup (knob lat) for a chain count that leaves the last workgroup partly filled;
    kernel_normal with thinning and a continuation, kernel_adapt."""
    from fmcmc_amd import _abi as abi
    set_knob(monkeypatch, "lat", lat)
    X, y = synth_linreg(n, p, 4100 + n + p)
    init = jitter_init([0.0] * (p + 1) + [float(np.std(y))], 7, 50 + p)
    init[:, -1] = np.abs(init[:, -1])
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, p + 2, init, nsteps=70, burnin=4, thin=3, calls=2, scale=0.03)
    assert abi.last_kernel() == "lat" + lat
    run_both(E, O, O.FAM_LINREG, X, y, O.K_UNIF_REFLECTIVE, p + 2, init, nsteps=50, min_=-0.04, max_=0.05,
             lb=[-9.0] * (p + 1) + [0.05], ub=9.0, fixed=[False, True] + [False] * p)
    assert abi.last_kernel() == "lat" + lat
    run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, p + 2, init, nsteps=70, calls=2, warmup=15)
    assert abi.last_kernel() == "spec-lat" + lat


def test_latency_form_iid_normal(E, O):
    """The iid Normal family (R/mcmc.R:141-144's example) is the linear model with an intercept and no covariate: with a handful
    of chains it runs on the latency form too (mh_sweep_lat<KIND, 0, 20>), every slot count, the reflective and uniform kernels."""
    from fmcmc_amd import _abi as abi
    rng = np.random.default_rng(12)
    for n in (700, 5000, 10240):
        y = 1.5 + 2.0 * rng.standard_normal(n)
        init = jitter_init([1.0, 2.0], 7, 5)
        init[:, -1] = np.abs(init[:, -1]) + 0.1
        run_both(E, O, O.FAM_IID_NORMAL, None, y, O.K_NORMAL, 2, init, nsteps=90, burnin=4, thin=3, calls=2, scale=0.03)
        assert abi.last_kernel() == "lat1"
        run_both(E, O, O.FAM_IID_NORMAL, None, y, O.K_UNIF_REFLECTIVE, 2, init, nsteps=60, min_=-0.04, max_=0.05, lb=[-40.0, 0.05], ub=40.0)
        assert abi.last_kernel() == "lat1"


def test_latency_form_step_windows(E, O, monkeypatch):
    """A long call of the normal kernels runs as step windows (bounded stream) in the latency form too: 96-step windows, the
    bits of the oracle's one call."""
    from fmcmc_amd import _abi as abi
    set_knob(monkeypatch, "window", "96")
    X, y = synth_linreg(3000, 3, 77)
    init = jitter_init([0, 0, 0, 0, float(np.std(y))], 5, 3)
    init[:, -1] = np.abs(init[:, -1])
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, init, nsteps=333, burnin=40, thin=7, calls=2, scale=0.03)
    assert abi.last_kernel() == "lat1"


# ---- BASELINE configs[1], [2] and [4] at EXACTLY their per-GPU width against the oracle (the oracle on 16 host threads)
def test_c2_exact_shape_equals_the_oracle(E, O):
    """configs[1]: 1024 chains x n = 10,000, k = 5, kernel_normal(scale = 0.02) -- the headline kernel (mh_sweep_mfma<1, 1, 20>,
    256 workgroups of four chains), 48 steps, every output and the carried state, bit for bit."""
    import bench
    from fmcmc_amd import _abi as abi
    cfg = bench.Config("c2")
    X, y, init = cfg.workload(cfg.chains, 0)
    assert X.shape == (10000, 3) and init.shape == (1024, 5)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, init, nsteps=48, seed=bench.CHAIN_SEED, scale=0.02, threads=16)
    assert abi.last_kernel() == "mfma"


def test_c3_exact_shape_equals_the_oracle(E, O):
    """configs[2]: 1024 chains, kernel_adapt() with its default warmup = 500 (R/kernel_adapt.R:118-160), 720 steps: the
    recursive mean / covariance, the Cholesky factor and the proposals of 220 adapting steps at full width, then a second
    call of 64 steps that continues the adapted state -- every output, Sigma, the running mean, abs_iter, bit for bit."""
    import bench
    from fmcmc_amd import _abi as abi
    cfg = bench.Config("c3")
    X, y, init = cfg.workload(cfg.chains, 0)
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, 5, init, nsteps=720, seed=bench.CHAIN_SEED, warmup=500, threads=16)
    assert abi.last_kernel() == "spec"
    st = rg.state if hasattr(rg, "state") else None
    del rg, ro, st


def test_c5_full_width_equals_the_oracle(E, O, monkeypatch):
    """configs[4] at its per-GPU width: 1024 chains, logistic n = 100,000, k = 6, kernel_normal_reflective(scale = .01, lb = -5,
    ub = 5), thin 10 -- the observation-sharded form with TWO chains per thread (logit_shard<5, 2, 2>, the instantiation the
    bench runs), 21 steps (two kept rows) and a continuation of 11, bit for bit."""
    import bench
    from fmcmc_amd import _abi as abi
    cfg = bench.Config("c5")
    X, y, init = cfg.workload(cfg.chains, 0)
    assert X.shape == (100000, 5) and init.shape == (1024, 6)
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL_REFLECTIVE, 6, init, nsteps=21, thin=10, seed=bench.CHAIN_SEED,
             prior_div=8.0, guard=False, scale=0.01, lb=-5.0, ub=5.0, threads=16)
    assert abi.last_kernel() == "logistic-shadow"        # (round 5: the owners in the shadow of the hand-overs, mh_sweep_logit2)
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL_REFLECTIVE, 6, init, nsteps=11, thin=1, seed=bench.CHAIN_SEED,
             prior_div=8.0, guard=False, scale=0.01, lb=-5.0, ub=5.0, threads=16)
    assert abi.last_kernel() == "logistic-shadow"
    set_knob(monkeypatch, "shadow", "0")                 # the same call on the general kernel's observation-sharded form
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL_REFLECTIVE, 6, init, nsteps=11, thin=2, seed=bench.CHAIN_SEED,
             prior_div=8.0, guard=False, scale=0.01, lb=-5.0, ub=5.0, threads=16)
    assert abi.last_kernel() == "logistic-sharded"


# ---------------------------------------------------------------------------------------------------------------------
# uniform kernels, the remaining update schemes, kernel_ram freq / constr
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind_name,C,n,p", [("unif", 5, 777, 3), ("unif_reflective", 64, 2048, 3),
                                              ("unif", 9, 10000, 3), ("unif_reflective", 6, 10000, 3),   # MFMA path
                                              ("unif", 3, 1000, 1)])                                       # specialised path
def test_uniform_kernels(E, O, kind_name, C, n, p):
    """R/kernel_unif.R:42-170: all three code paths (streamed, MFMA, wave-specialised) draw the same U(0,1) stream."""
    X, y = synth_linreg(n, p, 23 + n)
    init = jitter_init([0.0] * (p + 1) + [float(np.std(y))], C, 6)
    kind = O.K_UNIF if kind_name == "unif" else O.K_UNIF_REFLECTIVE
    kw = dict(min_=-0.04, max_=0.05)
    if kind == O.K_UNIF_REFLECTIVE:
        kw.update(lb=[-10.0] * (p + 1) + [0.0], ub=[10.0] * (p + 1) + [float(np.std(y)) * 1.02])
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, kind, p + 2, init, nsteps=260, calls=2, **kw)
    inc = ro.draws_cks[:, :, 1:] - ro.samples_cks[:, :, :-1]
    if kind == O.K_UNIF:
        assert inc.min() >= -0.04 - 1e-12 and inc.max() <= 0.05 + 1e-12 and 0.05 < ro.accept_count.mean() / 259 < 0.999


@pytest.mark.parametrize("scheme", ["ordered", "random", [3, 1, 4, 2, 5], "joint"])
@pytest.mark.parametrize("kind_name", ["normal_reflective", "unif"])
def test_update_schemes(E, O, scheme, kind_name):
    """plan_update_sequence (R/kernel.R:66-133): one parameter per step for every scheme but "joint"; two calls, the
    plan of scheme = "random" is the same in both (it belongs to the kernel object)."""
    X, y = synth_linreg(1200, 3, 41)
    init = jitter_init([0, 0, 0, 0, 4.0], 11, 2)
    if kind_name == "unif":
        kind, kw = O.K_UNIF, dict(min_=-0.1, max_=0.12)
    else:
        kind, kw = O.K_NORMAL_REFLECTIVE, dict(scale=0.06, lb=[-9, -9, -9, -9, 0.5], ub=9.0)
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, kind, 5, init, nsteps=200, calls=2, chain_base=7, scheme=scheme, **kw)
    moved = (ro.draws_cks[:, :, 1:] != ro.samples_cks[:, :, :-1]).sum(axis=1)      # parameters proposed per step
    assert np.all(moved <= (5 if scheme == "joint" else 1))
    if isinstance(scheme, list):   # step i updates scheme[(i - 1) mod 5]
        for i in range(2, 40):
            col = scheme[(i - 1) % 5] - 1
            others = [j for j in range(5) if j != col]
            assert np.all(ro.draws_cks[:, others, i - 1] == ro.samples_cks[:, others, i - 2])


@pytest.mark.parametrize("scheme", ["ordered", "random", [5, 1, 3, 2, 4]])
@pytest.mark.parametrize("chains,n,p", [(2, 1000, 3), (300, 1700, 2), (700, 600, 1), (1000, 2048, 3)])
def test_update_schemes_on_the_latency_form(E, O, scheme, chains, n, p):
    """Round 5: the single-parameter schemes of the normal / uniform kernels (R/kernel.R:66-133) on mh_sweep_lat's candidate wave -- one
    to FOUR chains per workgroup; "random" draws its plan in the kernel (Philox, the call's loop step) and hands it back.  Unbounded and
    reflective kernel, a fixed parameter, two calls with thinning: the oracle's bits and the oracle's plan."""
    from fmcmc_amd import _abi as abi
    X, y = synth_linreg(n, p, 600 + n)
    k = p + 2
    if isinstance(scheme, list):
        scheme = [c for c in scheme if c <= k]
    init = jitter_init([0.0] * (p + 1) + [float(np.std(y))], chains, 9)
    steps = 120 if chains < 100 else 40
    want = "lat%d" % min(4, (chains + 255) // 256)
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, k, init, nsteps=steps, calls=2, burnin=3, thin=2, scale=0.05, scheme=scheme)
    assert abi.last_kernel() == want, abi.last_kernel()
    lb = [-9.0] * (k - 1) + [0.5]
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL_REFLECTIVE, k, init, nsteps=steps, scale=0.3, lb=lb, ub=9.0, scheme=scheme)
    assert abi.last_kernel() == want
    if not isinstance(scheme, list):
        fixed = [False, True] + [False] * (k - 2)
        run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, k, init, nsteps=steps, scale=0.05, fixed=fixed, scheme=scheme)
        assert abi.last_kernel() == want
    if scheme == "random":   # a single free parameter at position j: R samples from 1:j
        fixed = [True] * k; fixed[k - 2] = False
        run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, k, init, nsteps=steps, scale=0.05, fixed=fixed, scheme=scheme)
        assert abi.last_kernel() == want
    # the logistic family: the same candidate wave on mh_sweep_lat<.., LOGISTIC>
    rng = np.random.default_rng(n + p)
    yl = (rng.uniform(size=n) < 1 / (1 + np.exp(-(0.3 + X @ np.linspace(0.5, -0.5, p))))).astype(np.float64)
    kl = p + 1
    if isinstance(scheme, list):
        scheme = [c for c in scheme if c <= kl]
    il = jitter_init([0.1] * kl, chains, 4)
    run_both(E, O, O.FAM_LOGISTIC, X, yl, O.K_NORMAL, kl, il, nsteps=steps, calls=2, burnin=3, thin=2, scale=0.1, prior_div=8.0, scheme=scheme)
    assert abi.last_kernel() == want.replace("lat", "lat-logit"), abi.last_kernel()
    run_both(E, O, O.FAM_LOGISTIC, X, yl, O.K_NORMAL_REFLECTIVE, kl, il, nsteps=steps, scale=0.3, lb=-0.7, ub=0.9, scheme=scheme)
    assert abi.last_kernel() == want.replace("lat", "lat-logit")
    if not isinstance(scheme, list):   # a fixed parameter, under the scheme and under the joint update
        fixed = [False, True] + [False] * (kl - 2)
        run_both(E, O, O.FAM_LOGISTIC, X, yl, O.K_NORMAL, kl, il, nsteps=steps, scale=0.1, prior_div=8.0, fixed=fixed, scheme=scheme)
        assert abi.last_kernel() == want.replace("lat", "lat-logit")
        run_both(E, O, O.FAM_LOGISTIC, X, yl, O.K_NORMAL, kl, il, nsteps=steps, scale=0.1, fixed=fixed)
        assert abi.last_kernel() == want.replace("lat", "lat-logit")


def test_update_schemes_with_fixed_parameters(E, O):
    X, y = synth_linreg(900, 3, 5)
    init = jitter_init([0, 0, 0, 0, 4.0], 4, 3)
    fixed = [False, True, False, True, False]
    for scheme in ("ordered", "random", [5, 1, 3]):
        rg, ro = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, init, nsteps=150, scale=0.05, fixed=fixed, scheme=scheme)
        assert np.all(ro.samples_cks[:, [1, 3], :] == init[:, [1, 3], None])
    # a single free parameter at position j: sample(j, nsteps, TRUE) draws from 1:j in R (R/kernel.R:110)
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, init, nsteps=150, scale=0.05,
                      fixed=[True, True, False, True, True], scheme="random")
    assert set(np.unique(rg.samples.cpu().numpy()[0, 3])) == {init[0, 3]}            # positions > j never move


def test_random_scheme_fed_replay_of_the_R_plan(E, O):
    """FED mode: logu, z and the update plan all come from the restated R generator in R's own order; the GPU then equals
    the oracle run on that R stream with canonical arithmetic, bit for bit."""
    import torch
    X, y = synth_linreg(600, 1, 8)
    init = np.array([[0.0, 0.0, 2.0]] * 2)
    om = O.Model(O.FAM_LINREG, X, y)
    ok = O.Kernel(O.K_NORMAL, 3, scale=0.1, scheme="random")
    ost = O.ChainState(init, 3)
    ro = O.run(om, ok, nsteps=300, seed=0, state=ost, rng_mode=O.RNG_RMT, math_mode=O.MATH_CANON, rng=O.RRng(77))
    g = O.RRng(77)                         # replay: per chain runif(nsteps), sample(), then one rnorm per step
    logu, z, cols = np.zeros((2, 300)), np.zeros((2, 300, 1)), np.zeros((2, 300), dtype=np.int32)
    for c in range(2):
        logu[c] = np.log(g.runif(300))
        cols[c] = g.sample_int(3, 300) - 1
        z[c, 1:, 0] = g.rnorm(299)
    assert np.array_equal(cols, ost.scheme_cols)
    gm = E.DeviceModel(O.FAM_LINREG, X, y)
    gk = E.KernelSpec(O.K_NORMAL, 3, ok.mu, ok.scale, ok.lb, ok.ub, ok.fixed, scheme=ok.scheme)
    gst = E.ChainState(init, 3)
    gst.scheme_cols = torch.as_tensor(cols).cuda()
    rg = E.sweep(gm, gk, gst, 300, fed_logu=torch.as_tensor(logu).cuda(), fed_z=torch.as_tensor(z).cuda())
    assert _bits_equal(rg.samples.cpu().numpy(), ro.samples_cks) and _bits_equal(rg.logpost.cpu().numpy(), ro.logpost)


@pytest.mark.parametrize("C,n,p,freq,constr", [(6, 900, 3, 3, None), (5, 10000, 3, 2, None), (4, 900, 3, 1, "band"),
                                                (3, 10000, 3, 4, "band")])
def test_ram_freq_and_constr(E, O, C, n, p, freq, constr):
    """R/kernel_ram.R:129 (`!(env$i %% freq)`) and :149-150 (constr mask), streamed and wave-specialised kernels."""
    X, y = synth_linreg(n, p, 77)
    k = p + 2
    init = jitter_init([0.0] * (p + 1) + [float(np.std(y))], C, 9)
    M = None
    if constr == "band":
        M = (np.abs(np.subtract.outer(np.arange(k), np.arange(k))) <= 1).astype(float)
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=220, calls=2, freq=freq, constr=M)
    if M is not None:
        assert np.all(ro.state.Sigma[:, M == 0] == 0) and np.any(ro.state.Sigma[:, 1, 0] != 0)


@pytest.mark.parametrize("C,n,p", [(6, 900, 3), (5, 10000, 3), (70, 2600, 30)])
@pytest.mark.parametrize("fam", [dict(ram_qfun=1), dict(ram_qfun=2, ram_df=3.0), dict(ram_qfun=2, ram_df=0.7, ram_eta_exp=0.9),
                                 dict(ram_eta_exp=0.51)])
def test_ram_qfun_and_eta_families(E, O, C, n, p, fam):
    """The built-in alternatives to kernel_ram's `qfun` and `eta` (R/kernel_ram.R:67-68, fmcmc_kernel.ram_*): normal and
    t(df) variates, eta = min(1, k i^-g); streamed, wave-specialised and (wide model) dataflow kernels."""
    X, y = synth_linreg(n, p, 78, beta=np.linspace(1.0, -1.0, p + 1))
    k = p + 2
    init = jitter_init([0.0] * (p + 1) + [float(np.std(y)) + 1.0], C, 10)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=160, calls=2, **fam)


@pytest.mark.parametrize("n,p", [(1, 0), (37, 1), (511, 2), (512, 3), (513, 3), (1025, 2), (3000, 0), (5000, 3), (8192, 3),
                                 (9728, 3), (9729, 1), (10240, 2),                      # one group of K = 4: p <= 3
                                 (700, 4), (2049, 5), (4096, 6), (5120, 7), (5119, 7)])  # two chained groups: p <= 7
def test_mfma_path_is_general_in_n_and_p(E, O, n, p):
    """mh_sweep_mfma with run-time n (batches of 2 observation slots, masks only in the last one) and run-time p (unused
    k-slots of the 4x4x4 blocks are zero): same bits as the oracle for every shape the 80 operand registers can hold."""
    X, y = synth_linreg(n, p, 1000 + n + p, beta=np.linspace(1.0, -1.0, p + 1))
    C = 6
    init = jitter_init([0.0] * (p + 1) + [float(np.std(y)) + 1.0], C, n)
    run_both(E, O, O.FAM_LINREG, X if p else None, y, O.K_NORMAL, p + 2, init, nsteps=130, burnin=10, thin=3, calls=2, scale=0.05)


def test_mfma_general_shapes_variants(E, O):
    X, y = synth_linreg(6000, 2, 9)
    init = jitter_init([0.0, 0.0, 4.0], 5, 3)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL_REFLECTIVE, 3, init, nsteps=150, intercept=False, scale=0.04,
             lb=[-3, -3, 0.5], ub=[3, 3, 6.0])
    init = jitter_init([0.0, 0.0, 0.0, 4.0], 7, 4)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_UNIF, 4, init, nsteps=150, min_=-0.03, max_=0.04, fixed=[False, True, False, False])
    X5, y5 = synth_linreg(3000, 5, 10, beta=[1, 2, -1, 0.5, 0.25, -2])
    init = jitter_init([0.0] * 6 + [4.0], 9, 5)
    run_both(E, O, O.FAM_LINREG, X5, y5, O.K_NORMAL, 7, init, nsteps=150, guard=False, scale=0.03)


@pytest.mark.parametrize("C,n,p,bw,freq,warmup", [(5, 900, 3, 0, 3, 20), (4, 10000, 3, 0, 2, 30), (6, 700, 2, 25, 1, 40),
                                                   (3, 10000, 3, 12, 1, 12), (4, 600, 1, 8, 5, 20)])
def test_adapt_window_and_stride(E, O, C, n, p, bw, freq, warmup):
    """kernel_adapt(bw > 0) / (freq > 1) (R/kernel_adapt.R:117-160): ring of the last rows per chain in HBM, same bits as
    the oracle's canonical restatement; a second call on the warmed-up kernel raises the reference's subscript error as a
    per-chain status when the rows it needs precede the call."""
    X, y = synth_linreg(n, p, 55 + n)
    k = p + 2
    init = jitter_init([0.0] * (p + 1) + [float(np.std(y))], C, 2)
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, k, init, nsteps=180, bw=bw, freq=freq, warmup=warmup, Sd=0.8)
    assert np.all(ro.status == 0) and ro.accept_count.min() > 0
    from fmcmc_amd import _abi as abi
    if bw == 0 and 2 <= freq <= 8 and 1 <= p <= 7 and n <= 512 * (20 if p <= 3 else 10 if p <= 5 else 8):
        assert abi.last_kernel() == "spec"      # (round 5: the last `freq` rows in an LDS ring of the register owner)
    if bw > 0 or freq > 2:
        om, ok = O.Model(O.FAM_LINREG, X, y), O.Kernel(O.K_ADAPT, k, bw=bw, freq=freq, warmup=warmup, Sd=0.8)
        gm = E.DeviceModel(O.FAM_LINREG, X, y)
        gk = E.KernelSpec(O.K_ADAPT, k, ok.mu, ok.scale, ok.lb, ok.ub, ok.fixed, freq=freq, warmup=warmup, bw=bw, Sd=0.8)
        ost, gst = O.ChainState(init, ok.kf), E.ChainState(init, ok.kf)
        for st, runner in ((ost, lambda: O.run(om, ok, nsteps=60, seed=3, state=ost)),
                           (gst, lambda: E.sweep(gm, gk, gst, 60, seed=3, check=False))):
            runner()
        r2o = O.run(om, ok, nsteps=60, seed=3, state=ost)
        r2g = E.sweep(gm, gk, gst, 60, seed=3, check=False)
        assert np.all(r2o.status == 4) and np.array_equal(r2g.status.cpu().numpy(), r2o.status)
        assert np.array_equal(r2g.status_step.cpu().numpy(), r2o.status_step)
        with pytest.raises(RuntimeError, match="subscript out of bounds"):
            E.raise_on_chain_error(r2g)


@pytest.mark.parametrize("freq,warmup", [(2, 0), (4, 9), (8, 30), (7, 3)])
@pytest.mark.parametrize("chains", [3, 700])
def test_adapt_stride_on_the_register_owner_and_its_latency_forms(E, O, freq, warmup, chains):
    """kernel_adapt(freq = 2 .. 8) (R/kernel_adapt.R:127-160: the last `freq` rows folded in together every freq-th step) on
    mh_sweep_spec's register owner -- LDS ring of the chain's last rows, the factor formed again only when Sigma has moved -- for the
    linear and the logistic model, one to three chains per workgroup and two calls (the second one's first update asks for rows
    that precede the call where freq allows it: status 4, as the reference's subscript error): the oracle's bits."""
    from fmcmc_amd import _abi as abi
    X, y = synth_linreg(2100, 3, 77 + freq)
    init = jitter_init([0.0] * 4 + [float(np.std(y))], chains, 5)
    steps = 90 if chains < 100 else 40
    run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, 5, init, nsteps=steps, freq=freq, warmup=warmup, calls=2)
    assert abi.last_kernel().startswith("spec"), abi.last_kernel()
    rng = np.random.default_rng(freq)
    Xl = rng.standard_normal((900, 2)); yl = (rng.uniform(size=900) < 0.4).astype(np.float64)
    initl = 0.1 * rng.standard_normal((chains, 3))
    run_both(E, O, O.FAM_LOGISTIC, Xl, yl, O.K_ADAPT, 3, initl, nsteps=steps, freq=freq, warmup=warmup, calls=2, prior_div=8.0)
    assert abi.last_kernel().startswith("spec-logit"), abi.last_kernel()


@pytest.mark.parametrize("kind_name", ["nmirror", "umirror"])
@pytest.mark.parametrize("C,n,p,scheme,fixed", [(5, 800, 2, "joint", False), (4, 10000, 3, "joint", False),
                                                 (6, 600, 2, "ordered", [False, True, False, False]),
                                                 (3, 900, 1, "random", False), (4, 1700, 1, "joint", [True, False, False])])
def test_mirror_kernels(E, O, kind_name, C, n, p, scheme, fixed):
    """R/kernel_mirror.R: running-mean mirror point, one-off tan() rescaling at abs_iter == nadapt, state carried into a
    second call (mu, scale, abs_iter, obs_arate)."""
    X, y = synth_linreg(n, p, 91 + n)
    k = p + 2
    base = [0.5] * (p + 1) + [float(np.std(y))]
    init = jitter_init(base, C, 8)
    kind = O.K_NMIRROR if kind_name == "nmirror" else O.K_UMIRROR
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, kind, k, init, nsteps=160, calls=2, mu=base, scale=0.15, warmup=120, nadapt=6,
                      lb=[-30.0] * (p + 1) + [0.05], ub=30.0, scheme=scheme, fixed=fixed)
    q = ro.state.obs_arate                                  # [C][k]: the one-off rate, then R's element-wise running mean through warm-up
    assert q.shape == (C, k) and np.all(np.isfinite(q)) and np.all((q >= 0) & (q <= 1))
    fx = np.asarray(fixed if fixed is not False else [False] * k, dtype=bool)
    if fx.any():                                            # (a fixed parameter never moves: its entry only decays from the one-off rate)
        assert np.all(q[:, fx] <= q[:, ~fx].max(axis=1, keepdims=True) + 1e-12)
    assert np.all(ro.state.abs_iter == 318)


@pytest.mark.parametrize("kind_name", ["nmirror", "umirror"])
@pytest.mark.parametrize("n", [800, 10000])
def test_mirror_obs_arate_through_warmup_and_across_calls(E, O, kind_name, n):
    """R/kernel_mirror.R:108-118, :246-253: after the one-off adaptation obs_arate keeps being updated through the warm-up --
    mean_recursive(as.double(ans[i-1, ] != ans[i-2, ]), obs_arate, abs_iter), element-wise: the closure's scalar turns into a
    k-vector -- and that is the kernel state a second call continues (R/kernel.R:405-422 writes it back).  (a) one call that ends
    inside the warm-up: k distinct running means; (b) a second call continuing INSIDE the warm-up: its first proposal (i = 2)
    has no ans[i-2, ] -- numeric(0) in R, which every later update keeps: NaN in every entry here; (c) a chain whose warm-up
    ends in the first call keeps its vector through the second.  General kernel (n = 800) and the MFMA owner (n = 10,000),
    the oracle's bits (inside run_both: outputs, mu, scale, obs_arate, abs_iter)."""
    X, y = synth_linreg(n, 2, 91 + n)
    k = 4
    base = [0.5, 0.5, 0.5, float(np.std(y))]
    init = jitter_init(base, 5, 8)
    kind = O.K_NMIRROR if kind_name == "nmirror" else O.K_UMIRROR
    common = dict(mu=base, scale=0.15, nadapt=6, lb=[-30.0] * 3 + [0.05], ub=30.0)
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, kind, k, init, nsteps=90, warmup=400, **common)              # (a)
    q = ro.state.obs_arate
    assert np.all(np.isfinite(q)) and np.all((q > 0) & (q < 1)) and len({tuple(r) for r in q.round(12)}) > 1
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, kind, k, init, nsteps=90, calls=2, warmup=400, **common)     # (b)
    assert np.all(np.isnan(ro.state.obs_arate)) and np.all(ro.state.abs_iter == 178)
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, kind, k, init, nsteps=90, calls=2, warmup=60, **common)      # (c)
    assert np.all(np.isfinite(ro.state.obs_arate))


@pytest.mark.parametrize("kind_name", ["nmirror", "umirror"])
@pytest.mark.parametrize("n,p,intercept", [(10000, 3, True), (6145, 1, True), (20001, 2, False), (5000, 5, True), (3073, 7, True), (4000, 9, True), (2000, 13, True),
                                           (1500, 3, True), (513, 2, True), (700, 9, True), (3072, 5, False)])
def test_mirror_kernels_on_the_streamed_mfma_evaluation(E, O, monkeypatch, kind_name, n, p, intercept):
    """kernel_nmirror / kernel_umirror with the joint scheme and no fixed parameter from 513 observations on (one resident slot for
    short data):
    mh_sweep_mfma_ad<KIND, NG, -2>, their owner between the barriers of the streamed MFMA evaluation (round 4; the all-family
    kernel took 17.8 us per step at C2's shape).  Warm-up mean, the one-off tan() rescaling, bounds, two calls with the state
    carried: the oracle's bits."""
    from fmcmc_amd import _abi as abi
    set_knob(monkeypatch, "shard", "0")
    X, y = synth_linreg(n, p, 93 + n, beta=np.linspace(0.6, -0.6, p + 1))
    k = p + 1 + (1 if intercept else 0)
    base = ([0.5] if intercept else []) + [0.5] * p + [float(np.std(y))]
    init = jitter_init(base, 6, 8)
    kind = O.K_NMIRROR if kind_name == "nmirror" else O.K_UMIRROR
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, kind, k, init, nsteps=150, calls=2, mu=base, scale=0.15, warmup=110, nadapt=6,
                      lb=[-30.0] * (k - 1) + [0.05], ub=30.0, intercept=intercept, burnin=3, thin=2)
    in_spec = (p <= 7 and n <= 512 * (20 if p <= 3 else (10 if p <= 5 else 8))) or (p <= 14 and n <= 2048)     # (round 5: there their owner runs on mh_sweep_spec)
    assert abi.last_kernel() == ("spec" if in_spec else "mfma-adaptive")
    assert np.all(np.isfinite(ro.state.obs_arate)) and np.all(ro.state.abs_iter == 298)
    if in_spec:
        set_knob(monkeypatch, "specmirror", "0")
        rg, ro = run_both(E, O, O.FAM_LINREG, X, y, kind, k, init, nsteps=150, calls=2, mu=base, scale=0.15, warmup=110, nadapt=6,
                          lb=[-30.0] * (k - 1) + [0.05], ub=30.0, intercept=intercept, burnin=3, thin=2)
        assert abi.last_kernel() == "mfma-adaptive"


@pytest.mark.parametrize("kind_name", ["nmirror", "umirror"])
@pytest.mark.parametrize("chains,n,p,intercept", [(1, 100, 3, True), (2, 1000, 3, True), (300, 100, 4, True), (700, 2500, 1, False),
                                                  (1001, 512, 7, True), (1030, 4096, 6, False), (5, 10240, 2, True)])
def test_mirror_kernels_on_the_wave_specialised_kernel_and_its_latency_forms(E, O, kind_name, chains, n, p, intercept):
    """Round 5: kernel_nmirror / kernel_umirror (R/kernel_mirror.R:66-131, :203-262; joint scheme, no fixed parameter) within
    mh_sweep_spec's registers -- their owner beside the evaluation on the kernel's LDS sequence words instead of between the barriers of
    the streamed MFMA evaluation, one to four chains per workgroup (up to 512 observations they ran on the general kernel: 2.7 us per
    step at the README's size).  Warm-up mean, the one-off tan() rescaling, the running rate, bounds, burn-in and thinning, two calls with
    the state carried, a ragged last workgroup: the oracle's bits (outputs, mu, scale, obs_arate, abs_iter)."""
    from fmcmc_amd import _abi as abi
    X, y = synth_linreg(n, p, 9300 + n + p, beta=np.linspace(0.6, -0.6, p + 1))
    k = p + 1 + (1 if intercept else 0)
    base = ([0.5] if intercept else []) + [0.5] * p + [float(np.std(y))]
    init = jitter_init(base, chains, 8)
    kind = O.K_NMIRROR if kind_name == "nmirror" else O.K_UMIRROR
    steps = 150 if chains < 100 else 40
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, kind, k, init, nsteps=steps, calls=2, mu=base, scale=0.15, warmup=steps - 40 + 10, nadapt=6,
                      lb=[-30.0] * (k - 1) + [0.05], ub=30.0, intercept=intercept, burnin=3, thin=2)
    assert abi.last_kernel() == ("spec" if chains > 768 else "spec-lat%d" % ((chains + 255) // 256)), abi.last_kernel()
    assert np.all(ro.state.abs_iter == 2 * steps - 2)


# ---------------------------------------------------------------------------------------------------------------------
# One-family instantiations of the streamed kernel: logistic (compile-time number of covariates, coefficients in SGPRs,
# chain-vectorised softplus), wide linear models (observation blocking), and the adaptive owners with a compile-time k
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cw", ["1", "2", "4"])
@pytest.mark.parametrize("p,intercept", [(0, True), (1, True), (3, False), (5, True), (6, True), (8, True), (9, True)])
def test_logistic_specialised_loops(E, O, monkeypatch, cw, p, intercept):
    """Every compile-time-p loop body of the logistic instantiations (p <= 28 / CW - 1 coefficients in SGPRs, beyond
    that the run-time loop), for 1, 2 and 4 chains per workgroup, normal and reflective kernels, ragged n, a chain count
    that is not a multiple of CW, plus large |eta| (both softplus tails)."""
    set_knob(monkeypatch, "cw", cw)
    rng = np.random.default_rng(100 + 10 * p + int(cw))
    n = 1500 + 37 * p
    X = rng.standard_normal((n, p)) * (3.0 if p == 5 else 1.0)        # p = 5: |eta| up to ~40
    k = p + (1 if intercept else 0)
    beta = rng.uniform(-1.5, 1.5, k)
    eta = (beta[0] if intercept else 0.0) + (X @ beta[(1 if intercept else 0):] if p else 0.0)
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-eta))).astype(np.float64)
    C = 2 * int(cw) + 1
    init = jitter_init(beta, C, 40 + p)
    run_both(E, O, O.FAM_LOGISTIC, X if p else None, y, O.K_NORMAL, k, init, nsteps=70, burnin=5, thin=2, calls=2,
             prior_div=8.0, scale=0.04, intercept=intercept)
    run_both(E, O, O.FAM_LOGISTIC, X if p else None, y, O.K_NORMAL_REFLECTIVE, k, init, nsteps=50, prior_div=0.0,
             scale=0.3, lb=-2.0, ub=2.0, intercept=intercept)


@pytest.mark.parametrize("p,intercept,C", [(1, True, 5), (3, False, 9), (5, True, 7), (8, True, 3), (5, True, 600), (2, True, 513),
                                           (9, True, 5), (12, False, 520), (16, True, 4), (13, True, 700), (11, False, 9)])
def test_logistic_observation_sharded(E, O, monkeypatch, p, intercept, C):
    """The observation-sharded logistic evaluation (logit_shard, knob shard=1 forces it at these sizes): 256 workgroups of two
    canonical lanes each evaluate ALL chains, thread = chain -- one chain per thread up to 512 chains, two side by side
    above; two observations per pass up to p = 8, one from p = 9 to 16 --, hand-pipelined fast loop and the checked form
    (large |eta|: beyond the table's 2400 rows), ragged n (a last
    slot that only some lanes hold), chain counts that leave workgroups without chains, two consecutive calls."""
    from fmcmc_amd import _abi as abi
    set_knob(monkeypatch, "shard", "1")
    rng = np.random.default_rng(900 + 10 * p + C)
    n = 512 * 3 + 37 * p + 1                                              # odd: the last slot holds lanes 0 .. 37 p only
    X = rng.standard_normal((n, p)) * (12.0 if p in (3, 11) else 1.0)     # p = 3, 11: |eta| up to ~60, off the table
    k = p + (1 if intercept else 0)
    beta = rng.uniform(-1.5, 1.5, k)
    eta = (beta[0] if intercept else 0.0) + X @ beta[(1 if intercept else 0):]
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-eta))).astype(np.float64)
    init = jitter_init(beta, C, 40 + p)
    steps = 24 if C > 100 else 60
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL, k, init, nsteps=steps, burnin=5, thin=2, calls=2,
             prior_div=8.0, scale=0.04, intercept=intercept)
    assert abi.last_kernel() == "logistic-shadow"
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL_REFLECTIVE, k, init, nsteps=steps, prior_div=0.0,
             scale=0.3, lb=-2.0, ub=2.0, intercept=intercept)
    assert abi.last_kernel() == "logistic-shadow"
    set_knob(monkeypatch, "shadow", "0")                 # (the general kernel's observation-sharded form, variates from the stream)
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL, k, init, nsteps=steps, burnin=5, thin=2, calls=2,
             prior_div=8.0, scale=0.04, intercept=intercept)
    assert abi.last_kernel() == "logistic-sharded"
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL_REFLECTIVE, k, init, nsteps=steps, prior_div=0.0,
             scale=0.3, lb=-2.0, ub=2.0, intercept=intercept)
    assert abi.last_kernel() == "logistic-sharded"


@pytest.mark.parametrize("n,p,intercept", [(100, 4, True), (1000, 2, False), (5000, 5, True), (4096, 7, True), (10240, 3, True), (513, 1, True)])
@pytest.mark.parametrize("chains", [2, 5, 600, 1100])
def test_logistic_on_the_wave_specialised_kernel_and_its_latency_forms(E, O, monkeypatch, n, p, intercept, chains):
    """Round 5: the logistic family on mh_sweep_spec (data in the compute lanes' registers, g table in LDS, register owners) -- the
    workflow vignette's own model, mcmc::logit's 100 observations and four covariates, first (vignettes/workflow-with-fmcmc.Rmd:22-60:
    kernel_normal, then kernel_adapt(freq = 1, warmup = 500)).  All four proposal kernels, one to four chains per workgroup, ragged n,
    with and without intercept, two calls: the oracle's bits."""
    from fmcmc_amd import _abi as abi
    if n * chains > 3_000_000:
        pytest.skip("oracle time")
    rng = np.random.default_rng(1000 * p + n + chains)
    X = rng.standard_normal((n, p))
    beta = np.linspace(0.6, -0.6, p + 1)
    eta = (beta[0] if intercept else 0.0) + X @ beta[1:]
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-eta))).astype(np.float64)
    k = p + (1 if intercept else 0)
    init = jitter_init(list(beta[(0 if intercept else 1):]), chains, 17 + p)
    steps = 60 if chains < 100 else 20
    kw = dict(nsteps=steps, calls=2, prior_div=8.0, intercept=intercept)
    # (the normal kernels with fewer than four chains per CU on short data: the latency form, mh_sweep_lat<.., LOGISTIC>)
    lat = chains <= 768 and n <= (6144 if chains <= 256 else 3072)
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL, k, init, scale=0.1, burnin=3, thin=2, **kw)
    assert abi.last_kernel().startswith("lat-logit" if lat else "spec-logit"), abi.last_kernel()
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL_REFLECTIVE, k, init, scale=0.3, lb=-0.7, ub=0.9, **dict(kw, prior_div=0.0))
    assert abi.last_kernel().startswith("lat-logit" if lat else "spec-logit"), abi.last_kernel()
    if lat:     # the same calls on the wave-specialised kernel (knob speclogit=2: no latency form)
        set_knob(monkeypatch, "speclogit", "2")
        run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL, k, init, scale=0.1, burnin=3, thin=2, **kw)
        assert abi.last_kernel().startswith("spec-logit"), abi.last_kernel()
        run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL_REFLECTIVE, k, init, scale=0.3, lb=-0.7, ub=0.9, **dict(kw, prior_div=0.0))
        assert abi.last_kernel().startswith("spec-logit")
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_ADAPT, k, init, warmup=6, **kw)
    assert abi.last_kernel().startswith("spec-logit")
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_RAM, k, init, **kw)
    assert abi.last_kernel().startswith("spec-logit")
    # the bounded kernel_ram: a second evaluation request in the steps in which the reflection moved the proposal (SpecSyncB)
    rg, ro = run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_RAM, k, np.clip(init, -0.65, 0.85), lb=-0.7, ub=0.9, **kw)
    assert abi.last_kernel().startswith("spec-logit") and ro.accept_count.sum() > 0


@pytest.mark.parametrize("n,p,intercept,chains", [(200, 8, True, 3), (50, 12, True, 300), (1, 9, False, 2), (2048, 15, True, 5), (1500, 10, False, 700), (513, 15, False, 1000)])
def test_logistic_with_eight_to_fifteen_covariates_on_the_latency_form(E, O, monkeypatch, n, p, intercept, chains):
    """Round 5: the logistic family with 8 .. 15 covariates (k <= 16) on up to 2048 observations under the normal / uniform kernels:
    mh_sweep_lat<KIND, P, 4, LOGISTIC>, one to four chains per workgroup (they ran on the general kernel, 3 - 4.5 us per step at n = 200).
    Joint and single-parameter schemes, a fixed parameter, reflective bounds, two calls with burn-in and thinning: the oracle's bits."""
    from fmcmc_amd import _abi as abi
    rng = np.random.default_rng(100 * p + n + chains)
    X = rng.standard_normal((n, p))
    beta = np.linspace(0.5, -0.5, p + 1)
    eta = (beta[0] if intercept else 0.0) + X @ beta[1:]
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-eta))).astype(np.float64)
    k = p + (1 if intercept else 0)
    init = jitter_init(list(beta[(0 if intercept else 1):]), chains, 17 + p)
    steps = 60 if chains < 100 else 24
    want = "lat-logit%d" % min(4, (chains + 255) // 256)
    kw = dict(nsteps=steps, intercept=intercept)
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL, k, init, scale=0.05, burnin=3, thin=2, calls=2, prior_div=8.0, **kw)
    assert abi.last_kernel() == want, abi.last_kernel()
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL_REFLECTIVE, k, np.clip(init, -0.65, 0.65), scale=0.2, lb=-0.7, ub=0.7, **kw)
    assert abi.last_kernel() == want
    fixed = [False] * k
    fixed[1] = True
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_UNIF, k, init, min_=-0.04, max_=0.05, fixed=fixed, prior_div=8.0, **kw)
    assert abi.last_kernel() == want
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL, k, init, scale=0.05, scheme="random", prior_div=8.0, **kw)
    assert abi.last_kernel() == want
    # kernel_adapt / kernel_ram: mh_sweep_spec<P, 4, KIND, LOGISTIC>, the register owner at the compile-time width k <= 16
    want_a = "spec-logit" if chains > 768 else "spec-logit-lat%d" % ((chains + 255) // 256)
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_ADAPT, k, init, warmup=6, calls=2, burnin=2, thin=3, prior_div=8.0, **kw)
    assert abi.last_kernel() == want_a, abi.last_kernel()
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_ADAPT, k, np.clip(init, -0.65, 0.65), warmup=4, lb=-0.7, ub=0.7, **kw)
    assert abi.last_kernel() == want_a
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_RAM, k, init, calls=2, prior_div=8.0, **kw)
    assert abi.last_kernel() == want_a
    set_knob(monkeypatch, "specwide", "0")
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_RAM, k, init, prior_div=8.0, **kw)
    assert not abi.last_kernel().startswith("spec-logit")
    set_knob(monkeypatch, "speclogit", "0")
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL, k, init, scale=0.05, prior_div=8.0, **kw)
    assert not abi.last_kernel().startswith("lat-logit")


def test_logistic_fixed_parameter_beyond_the_latency_kernel_s_slots(E, O):
    """Found by the randomised soak at the end of round 5 (test_randomised_option_cases[1566]): a fixed parameter under kernel_normal on the
    logistic family went to the latency form up to 12 slots per lane whatever the kernel holds -- p = 6 holds 8 (n <= 4096): at n = 4097 the
    launch ran no loop at all and returned zeros.  The oracle's bits at the slot-count edges of p = 4 .. 7."""
    from fmcmc_amd import _abi as abi
    for n, p in ((4097, 6), (4096, 6), (5121, 5), (5120, 4), (4097, 7), (6000, 3)):
        rng = np.random.default_rng(n + p)
        X = rng.standard_normal((n, p))
        beta = np.linspace(0.5, -0.5, p)
        y = (rng.uniform(size=n) < 1 / (1 + np.exp(-(X @ beta)))).astype(np.float64)
        init = jitter_init(list(beta), 2, 5)
        fixed = [True] + [False] * (p - 2) + [True]
        rg, ro = run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL, p, init, nsteps=70, burnin=1, calls=2, chain_base=3, intercept=False, prior_div=8.0,
                          fixed=fixed, scale=0.01)
        assert ro.accept_count.sum() > 0


@pytest.mark.parametrize("form", ["shadow", "spec"])
def test_logistic_round5_kernels_edge_cases(E, O, monkeypatch, form):
    """mh_sweep_logit2 ("logistic-shadow", forced by shard=1) and the logistic family on mh_sweep_spec ("spec-logit") at the edges: two
    and three steps, a burn-in that leaves one row, a thinning interval one short of the call, one chain, chain counts that leave the last
    workgroup ragged, 1025 chains (two launches of the sharded form), a continuation of a call, and chains whose log-posterior is NaN
    from the start (status 1 / 2 with the step and the vector): the oracle's bits."""
    import torch
    from fmcmc_amd import _abi as abi
    if torch.cuda.get_device_properties(0).multi_processor_count < 256:
        pytest.skip("the sharded form needs all 256 CUs")
    if form == "shadow":
        set_knob(monkeypatch, "shard", "1")
    want = "logistic-shadow" if form == "shadow" else ("spec-logit", "lat-logit")
    rng = np.random.default_rng(5150)
    n, p = 1500, 3
    X = rng.standard_normal((n, p)); beta = np.array([0.3, 0.7, -0.6, 0.4])
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-(beta[0] + X @ beta[1:])))).astype(np.float64)
    k = p + 1
    for chains, kw in ((1, dict(nsteps=2)), (3, dict(nsteps=3)), (5, dict(nsteps=40, burnin=39)), (7, dict(nsteps=33, thin=32)),
                       (9, dict(nsteps=65, burnin=1, thin=3, calls=3)), (1025, dict(nsteps=12, thin=5))):
        init = jitter_init(list(beta), chains, 3 + chains)
        run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL_REFLECTIVE, k, init, scale=0.2, lb=-1.0, ub=1.0, prior_div=8.0, guard=False, **kw)
        assert abi.last_kernel().startswith(want), (abi.last_kernel(), chains)
    # NaN from the start (an initial vector with a NaN coefficient): the chain stops at its first decision, the others run on
    init = jitter_init(list(beta), 6, 11)
    init[2, 1] = np.nan
    init[4, 0] = np.inf          # (inf - inf in the ratio)
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL, k, init, nsteps=20, scale=0.1, prior_div=0.0, guard=False)
    assert abi.last_kernel().startswith(want)


def test_logit_shard_priority_turns_change_no_bit(E, monkeypatch):
    """logit_shard gives the younger wave of every SIMD the issue priority for the first part of its passes and regulates where that
    turn ends from the two waves' finishing times (round 5): TIMING only.  The same call with no turn, with a fixed early and a fixed
    late turn and with the regulated one: identical samples, log-posteriors and states, on both observation-sharded kernels."""
    import torch
    from fmcmc_amd import _abi as abi
    if torch.cuda.get_device_properties(0).multi_processor_count < 256:
        pytest.skip("the sharded forms need all 256 CUs")
    set_knob(monkeypatch, "shard", "1")
    rng = np.random.default_rng(811)
    n, p, C = 512 * 9 + 77, 4, 700
    X = rng.standard_normal((n, p)); y = (rng.uniform(size=n) < 0.45).astype(np.float64)
    k = p + 1
    big = E.DBL_MAX
    gm = E.DeviceModel(abi.FAM_LOGISTIC, X, y, intercept=True, guard=False, prior_div=8.0)
    init = 0.1 * rng.standard_normal((C, k))
    for kind, want in ((abi.KERNEL_NORMAL, "logistic-shadow"), (abi.KERNEL_ADAPT, "logistic-shadow"), (abi.KERNEL_ADAPT, "logistic-sharded")):
        set_knob(monkeypatch, "shadow", "0" if want == "logistic-sharded" else "-1")       # (the general kernel's sharded form too)
        gk = (E.KernelSpec(kind, k, np.zeros(k), np.full(k, 0.05), np.full(k, -big), np.full(k, big), np.zeros(k, np.uint8)) if kind == abi.KERNEL_NORMAL
              else E.KernelSpec(kind, k, np.zeros(k), np.ones(k), np.full(k, -big), np.full(k, big), np.zeros(k, np.uint8), warmup=10))
        ref = None
        for turn in ("0", "10250", "10900", "-1"):
            set_knob(monkeypatch, "turn", turn)
            st = E.ChainState(init, gk.kf)
            r = E.sweep(gm, gk, st, 60, seed=77)
            assert abi.last_kernel() == want
            got = (r.samples.clone(), r.logpost.clone(), st.theta0.clone(), st.Sigma.clone())
            if ref is None:
                ref = got
            else:
                assert all(torch.equal(a, b) for a, b in zip(ref, got)), (want, turn)


@pytest.mark.parametrize("n", [100000, 99841, 100351])
@pytest.mark.parametrize("form", ["chain-sharded", "observation-sharded"])
def test_c5_exact_shape_equals_the_oracle(E, O, monkeypatch, n, form):
    """BASELINE config C5 at its real shape -- logistic, n = 100,000 (195 / 196 observations per canonical lane: ~98 trips of
    the two-operand-set loop and both of its tails; 99,841 / 100,351: the neighbouring tail cases), p = 5 + intercept,
    kernel_normal_reflective(scale = .01, lb = -5, ub = 5) -- against the ORACLE, bit for bit: 10 chains at four per
    workgroup, thin 10 and thin 1, on the chain-sharded loop (logit_partials<4, 5>) and on the observation-sharded one
    (logit_shard<5, 1>), the form the full config runs on."""
    import bench
    from fmcmc_amd import _abi as abi
    cfg = bench.Config("c5")
    X, y, init = cfg.workload(10, 0)
    X, y = X[:n], y[:n]
    if n > cfg.n:                                                       # (the config's generator, continued)
        rng = np.random.default_rng(77)
        Xe = rng.standard_normal((n - cfg.n, cfg.p))
        X, y = np.vstack([X, Xe]), np.concatenate([y, (rng.uniform(size=n - cfg.n) < 0.3).astype(np.float64)])
    set_knob(monkeypatch, "cw", "4")
    set_knob(monkeypatch, "shard", "1" if form == "observation-sharded" else "0")
    for thin, steps in ((10, 31), (1, 12)):
        run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL_REFLECTIVE, 6, init, nsteps=steps, thin=thin, prior_div=8.0,
                 guard=False, scale=0.01, lb=-5.0, ub=5.0)
        assert abi.last_kernel() == ("logistic-shadow" if form == "observation-sharded" else "streamed-logistic")


@pytest.mark.parametrize("cw", ["1", "2"])
@pytest.mark.parametrize("n,p", [(1023, 16), (2500, 21), (1024 * 3 + 1, 33)])
def test_wide_linreg_instantiations(E, O, monkeypatch, cw, n, p):
    """Wide linear models (p >= 16) run one-family / one-kernel instantiations with two observations per thread: odd
    observation counts (padded second slot), column remainders (p % 8), normal / reflective / RAM kernels."""
    set_knob(monkeypatch, "cw", cw)
    X, y = synth_linreg(n, p, 7000 + n + p, beta=np.linspace(1.0, -1.0, p + 1))
    C = 2 * int(cw) + 1
    init = jitter_init(list(np.linspace(1.0, -1.0, p + 1)) + [4.0], C, n + p)
    init[:, -1] = np.abs(init[:, -1])
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, p + 2, init, nsteps=60, burnin=3, thin=2, calls=2, scale=0.01)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL_REFLECTIVE, p + 2, init, nsteps=40, scale=0.2, lb=-3.0, ub=6.0)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, p + 2, init, nsteps=60, calls=2)


@pytest.mark.parametrize("kind_name", ["adapt", "ram"])
@pytest.mark.parametrize("n,p,intercept,fixed", [(10000, 3, True, None), (10000, 3, False, None), (1000, 1, True, None),
                                                 (1000, 1, False, None), (10000, 3, True, [False, True, False, False, False])])
def test_adaptive_owner_variants(E, O, kind_name, n, p, intercept, fixed):
    """Register-row adaptive owners of mh_sweep_spec: k = P + 2 and k = P + 1 as compile-time constants, and the generic
    k <= 8 / LDS versions (a fixed parameter), with continuation of the adapted state."""
    X, y = synth_linreg(n, p, 300 + n + p)
    k = p + 1 + (1 if intercept else 0)
    base = ([0.0] if intercept else []) + [0.0] * p + [float(np.std(y))]
    init = jitter_init(base, 6, 50 + k)
    init[:, -1] = np.abs(init[:, -1])
    kw = dict(intercept=intercept)
    if fixed is not None:
        kw["fixed"] = fixed
    if kind_name == "adapt":
        run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, k, init, nsteps=140, calls=2, warmup=30, **kw)
    else:
        run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=120, calls=2, **kw)


@pytest.mark.parametrize("n,p", [(600, 3), (5000, 3), (9000, 3), (9217, 3), (10240, 3), (3000, 2), (700, 1), (5120, 4), (4000, 5), (4096, 7), (2049, 6)])
def test_wave_specialised_kernel_at_any_shape(E, O, n, p):
    """mh_sweep_spec (kernel_adapt / kernel_ram) with its slot count a run-time choice among compute loops (round 4; it
    existed for n in (9728, 10240] at p = 3 and (512, 1024] at p = 1 only): even and odd slot counts (9217: 19 slots, the
    partly filled one is then the second to last of the 20 the loop walks), ragged n, every p the kernel takes."""
    from fmcmc_amd import _abi as abi
    X, y = synth_linreg(n, p, 4100 + n + p)
    k = p + 2
    init = jitter_init([0.0] * (p + 1) + [float(np.std(y))], 5, 90 + p)
    init[:, -1] = np.abs(init[:, -1])
    run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, k, init, nsteps=90, calls=2, warmup=20)
    assert abi.last_kernel() == "spec"
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=70, calls=2)
    assert abi.last_kernel() == "spec"


@pytest.mark.parametrize("n,p", [(10241, 3), (12000, 3), (20000, 3), (20481, 2), (50001, 3), (5121, 5), (9000, 7), (12345, 4)])
def test_mfma_kernel_beyond_its_operand_registers(E, O, monkeypatch, n, p):
    """mh_sweep_mfma<.., EXT> (round 4): 16 (p <= 3) or 8 (p <= 7) observation slots resident, the rest streamed every step
    from the operand-order copy -- the shapes that fell to the general kernel at observation 10,241 / 5,121.  Ragged last
    slot, one and many streamed slots, normal and reflective kernels, two consecutive calls."""
    set_knob(monkeypatch, "shard", "0")     # (few chains on long data would take the long-data form: test_few_chains_on_long_data)
    from fmcmc_amd import _abi as abi
    X, y = synth_linreg(n, p, 5200 + n + p)
    k = p + 2
    init = jitter_init([0.0] * (p + 1) + [float(np.std(y))], 6, 95 + p)
    init[:, -1] = np.abs(init[:, -1])
    steps = 40 if n > 20000 else 80
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, k, init, nsteps=steps, burnin=3, thin=2, calls=2, scale=0.02)
    assert abi.last_kernel() == "mfma-streamed"
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL_REFLECTIVE, k, init, nsteps=steps, scale=0.3, lb=-6.0, ub=9.0)
    assert abi.last_kernel() == "mfma-streamed"


@pytest.mark.parametrize("n,p,intercept", [(513, 8, True), (2048, 9, True), (2049, 11, False), (3000, 10, True), (10000, 11, True), (700, 12, True),
                                           (1536, 13, True), (1537, 14, True), (6000, 14, False), (10001, 15, False), (12345, 12, True), (4097, 8, False)])
def test_mfma_kernel_with_eight_to_fifteen_covariates(E, O, monkeypatch, n, p, intercept):
    """8 <= p <= 15 (k <= 16) on mh_sweep_mfma<.., EXT> with three / four operand groups per slot (round 4: these models ran on the
    general kernel, 4x the time per flop of p = 7): four / three slots resident, or one for short data, the rest streamed; ragged
    last slots, the slot-count edges, with and without intercept, normal / reflective / uniform kernels, a fixed parameter."""
    set_knob(monkeypatch, "shard", "0")     # (few chains on long data would take the long-data form: test_few_chains_on_long_data)
    from fmcmc_amd import _abi as abi
    X, y = synth_linreg(n, p, 5300 + n + p, beta=np.linspace(0.8, -0.8, p + 1))
    k = p + 1 + (1 if intercept else 0)
    init = jitter_init(([0.0] if intercept else []) + [0.0] * p + [float(np.std(y))], 5, 96 + p)
    init[:, -1] = np.abs(init[:, -1])
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, k, init, nsteps=60, burnin=3, thin=2, calls=2, scale=0.02, intercept=intercept)
    assert abi.last_kernel() == "mfma-streamed"
    fixed = [False] * k
    fixed[2] = True
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL_REFLECTIVE, k, init, nsteps=50, scale=0.3, lb=-6.0, ub=9.0, intercept=intercept, fixed=fixed)
    assert abi.last_kernel() == "mfma-streamed"
    run_both(E, O, O.FAM_LINREG, X, y, O.K_UNIF, k, init, nsteps=40, min_=-0.02, max_=0.03, intercept=intercept)
    assert abi.last_kernel() == "mfma-streamed"


@pytest.mark.parametrize("n,p,intercept", [(1, 8, True), (17, 9, False), (60, 12, True), (200, 10, True), (300, 15, False), (511, 13, True), (512, 8, True)])
def test_eight_to_fifteen_covariates_on_small_data(E, O, monkeypatch, n, p, intercept):
    """Round 5: 8 <= p <= 15 with up to 512 observations (a regression with ten covariates on two hundred observations) ran on the general
    kernel: 2.4 - 5 us per step under the normal kernels, 9 - 26 under kernel_adapt, where p = 7 takes 0.7 / 2.4.  The streamed MFMA forms
    (mh_sweep_mfma<.., EXT>, mh_sweep_mfma_ad) with nothing streamed: their one resident slot is the last, with its padding.  Every proposal
    kernel, n = 1 .. 512, with and without intercept, two calls: the oracle's bits; knob tinymfma=0: the old route."""
    from fmcmc_amd import _abi as abi
    X, y = synth_linreg(n, p, 8300 + n + p, beta=np.linspace(0.8, -0.8, p + 1))
    k = p + 1 + (1 if intercept else 0)
    sd = float(np.std(y)) if n > 1 else 1.0
    init = jitter_init(([0.0] if intercept else []) + [0.0] * p + [sd + 0.5], 6, 96 + p)
    init[:, -1] = np.abs(init[:, -1])
    kw = dict(intercept=intercept)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, k, init, nsteps=60, burnin=3, thin=2, calls=2, scale=0.02, **kw)
    assert abi.last_kernel() == "mfma-streamed", abi.last_kernel()
    fixed = [False] * k
    fixed[2] = True
    run_both(E, O, O.FAM_LINREG, X, y, O.K_UNIF_REFLECTIVE, k, init, nsteps=50, min_=-0.3, max_=0.3, lb=-6.0, ub=9.0, fixed=fixed, **kw)
    assert abi.last_kernel() == "mfma-streamed"
    wide = "spec" if p <= 14 else "mfma-adaptive"      # (8 .. 14 covariates: the wave-specialised kernel, test_adaptive_kernels_with_eight_to_fourteen_covariates...)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, k, init, nsteps=70, calls=2, warmup=15, **kw)
    assert abi.last_kernel() == wide, abi.last_kernel()
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=60, calls=2, burnin=2, thin=3, **kw)
    assert abi.last_kernel() == wide
    set_knob(monkeypatch, "specwide", "0")
    run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, k, init, nsteps=70, calls=2, warmup=15, **kw)
    assert abi.last_kernel() == "mfma-adaptive", abi.last_kernel()
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=60, calls=2, burnin=2, thin=3, **kw)
    assert abi.last_kernel() == "mfma-adaptive"
    base = list(init[0])
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NMIRROR, k, init, nsteps=90, calls=2, mu=base, scale=0.15, warmup=60, nadapt=6, lb=[-30.0] * (k - 1) + [0.05], ub=30.0, **kw)
    assert abi.last_kernel() == "mfma-adaptive"
    set_knob(monkeypatch, "tinymfma", "0")
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, k, init, nsteps=40, scale=0.02, **kw)
    assert abi.last_kernel() not in ("mfma-streamed", "mfma-adaptive")


@pytest.mark.parametrize("chains,n,p,intercept", [(2, 200, 8, True), (200, 50, 12, True), (300, 2048, 14, True), (1000, 1537, 15, False), (3, 1, 10, True)])
def test_eight_to_fifteen_covariates_on_the_latency_form(E, O, monkeypatch, chains, n, p, intercept):
    """Round 5: the normal / uniform kernels of the linear model with 8 .. 15 covariates (k <= 16) on up to 2048 observations on
    mh_sweep_lat<KIND, P, 4>: the joint scheme with one chain per compute unit (beyond, the streamed MFMA form), the single-parameter
    schemes with up to four (they ran on the general kernel).  Reflective bounds, a fixed parameter, two calls with burn-in and thinning:
    the oracle's bits and the oracle's plan."""
    from fmcmc_amd import _abi as abi
    X, y = synth_linreg(n, p, 8800 + n + p, beta=np.linspace(0.8, -0.8, p + 1))
    k = p + 1 + (1 if intercept else 0)
    sd = float(np.std(y)) if n > 1 else 1.0
    init = jitter_init(([0.0] if intercept else []) + [0.0] * p + [sd + 0.5], chains, 96 + p)
    init[:, -1] = np.abs(init[:, -1])
    per_cu = (chains + 255) // 256
    steps = 60 if chains < 100 else 24
    kw = dict(nsteps=steps, intercept=intercept)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, k, init, burnin=3, thin=2, calls=2, scale=0.02, **kw)
    assert abi.last_kernel() == ("lat1" if per_cu == 1 else "mfma-streamed"), abi.last_kernel()
    fixed = [False] * k
    fixed[2] = True
    run_both(E, O, O.FAM_LINREG, X, y, O.K_UNIF_REFLECTIVE, k, init, min_=-0.3, max_=0.3, lb=-6.0, ub=9.0, fixed=fixed, **kw)
    assert abi.last_kernel() == ("lat1" if per_cu == 1 else "mfma-streamed")
    for scheme in ("ordered", "random"):
        run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, k, init, scale=0.05, scheme=scheme, calls=2, **kw)
        assert abi.last_kernel() == "lat%d" % per_cu, abi.last_kernel()
    set_knob(monkeypatch, "specwide", "0")
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, k, init, scale=0.02, **kw)
    assert abi.last_kernel() == "mfma-streamed"


@pytest.mark.parametrize("chains,n,p,intercept,fixed_idx", [(2, 1000, 3, True, [1]), (300, 100, 3, True, [0, 2]), (700, 3000, 2, False, [2]), (1001, 777, 5, True, [6]),
                                                            (1030, 100, 6, False, [0, 1, 2, 3, 4, 5]), (6, 4000, 6, True, [3]), (5, 2000, 1, True, [0]),
                                                            (6, 12001, 3, True, [1]), (5, 6000, 6, False, [0, 6])])
def test_fixed_parameters_on_the_register_owner_and_its_latency_forms(E, O, monkeypatch, chains, n, p, intercept, fixed_idx):
    """Round 5: kernel_adapt / kernel_ram with FIXED parameters (R/kernel.R `fixed`, R/kernel_adapt.R:84-182 and R/kernel_ram.R:90-160 work in
    the space `which` spans) on mh_sweep_spec's register owner of run-time width: the free parameters on the first lanes, the fixed ones behind
    them as passengers (they ran on the owners with their matrices in LDS: 3.8 / 3.1 against 1.8 / 2.1 us per step at the README's size).  One
    or several fixed parameters -- the first, the last (sigma), all but one --, bounds, until, kernel_ram's freq / qfun, two calls with
    burn-in and thinning, step windows: the oracle's bits."""
    from fmcmc_amd import _abi as abi
    X, y = synth_linreg(n, p, 7700 + n + p, beta=np.linspace(0.7, -0.7, p + 1))
    k = p + 1 + (1 if intercept else 0)
    init = jitter_init(([0.1] if intercept else []) + [0.1] * p + [float(np.std(y))], chains, 77 + p)
    init[:, -1] = np.abs(init[:, -1]) + 0.2
    fixed = [j in fixed_idx for j in range(k)]
    steps = 90 if chains < 100 else 30
    want = "spec" if chains > 768 else "spec-lat%d" % ((chains + 255) // 256)
    if n > 512 * (20 if p <= 3 else (10 if p <= 5 else 8)):     # (beyond mh_sweep_spec's registers: the same owner between the barriers of mh_sweep_mfma_ad)
        want = "mfma-adaptive"
        set_knob(monkeypatch, "shard", "0")
    kw = dict(intercept=intercept, fixed=fixed)
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, k, init, nsteps=steps, calls=2, warmup=10, burnin=3, thin=2, **kw)
    assert abi.last_kernel() == want, abi.last_kernel()
    assert np.all(ro.samples_cks[:, fixed, :] == init[:, fixed, None])
    if want == "mfma-adaptive":      # (and the bounded kernel_ram there)
        run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=steps, lb=[-0.4] * (k - 1) + [0.3], ub=[0.4] * (k - 1) + [float(np.std(y)) + 0.5], **kw)
        assert abi.last_kernel() == want
    run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, k, init, nsteps=steps, warmup=5, lb=-4.0, ub=9.0, until=float(steps - 8), **kw)
    assert abi.last_kernel() == want
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=steps, calls=2, burnin=2, thin=3, **kw)
    assert abi.last_kernel() == want
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=steps, ram_qfun=1, warmup=4, freq=2, **kw)
    assert abi.last_kernel() == want
    set_knob(monkeypatch, "window", "16")
    run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, k, init, nsteps=steps, calls=2, warmup=10, **kw)
    assert abi.last_kernel() == want
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=steps, calls=2, **kw)
    assert abi.last_kernel() == want


@pytest.mark.parametrize("chains,n,p,intercept,fix", [(3, 200, 8, True, False), (300, 50, 12, True, False), (700, 2048, 14, True, False), (1001, 1537, 9, False, False),
                                                      (1030, 600, 13, False, False), (5, 1024, 11, True, True), (2, 1, 10, True, False)])
def test_adaptive_kernels_with_eight_to_fourteen_covariates_on_the_wave_specialised_kernel_and_its_latency_forms(E, O, monkeypatch, chains, n, p, intercept, fix):
    """Round 5: kernel_adapt / kernel_ram with 8 .. 14 covariates (k <= 16) on up to 2048 observations: mh_sweep_spec<P, 4, KIND> with the
    register owner at the compile-time width k = P + 2 / P + 1 -- in this kernel the owner waves hold no operands, so rows of 16 fit (the
    streamed MFMA evaluation with the owners' matrices in LDS took 4.3 - 9.5 us per step at n = 200).  One to four chains per workgroup, a ragged
    last workgroup, a fixed parameter (the LDS owners of the same kernel), warm-up / until, freq of kernel_ram, two calls with burn-in and
    thinning, step windows: the oracle's bits."""
    from fmcmc_amd import _abi as abi
    X, y = synth_linreg(n, p, 6600 + n + p, beta=np.linspace(0.7, -0.7, p + 1))
    k = p + 1 + (1 if intercept else 0)
    sd = float(np.std(y)) if n > 1 else 1.0
    init = jitter_init(([0.0] if intercept else []) + [0.0] * p + [sd + 0.3], chains, 98 + p)
    init[:, -1] = np.abs(init[:, -1])
    fixed = [False] * k
    if fix:
        fixed[1] = True
    steps = 80 if chains < 100 else 30
    want = "spec" if chains > 768 else "spec-lat%d" % ((chains + 255) // 256)
    kw = dict(intercept=intercept, fixed=fixed)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, k, init, nsteps=steps, calls=2, warmup=10, burnin=3, thin=2, **kw)
    assert abi.last_kernel() == want, abi.last_kernel()
    run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, k, init, nsteps=steps, warmup=5, lb=-4.0, ub=9.0, until=float(steps - 8), **kw)
    assert abi.last_kernel() == want
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=steps, calls=2, burnin=2, thin=3, **kw)
    assert abi.last_kernel() == want
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=steps, ram_qfun=1, warmup=4, freq=2, **kw)
    assert abi.last_kernel() == want
    if not fix:      # (the mirror kernels: joint scheme, no fixed parameter)
        run_both(E, O, O.FAM_LINREG, X, y, O.K_NMIRROR, k, init, nsteps=steps, calls=2, mu=list(init[0]), scale=0.1, warmup=steps - 10, nadapt=6,
                 lb=[-30.0] * (k - 1) + [0.05], ub=30.0, intercept=intercept)
        assert abi.last_kernel() == want
        run_both(E, O, O.FAM_LINREG, X, y, O.K_UMIRROR, k, init, nsteps=steps, mu=list(init[0]), scale=0.1, warmup=steps - 10, nadapt=6, intercept=intercept)
        assert abi.last_kernel() == want
    set_knob(monkeypatch, "window", "16")
    run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, k, init, nsteps=steps, calls=2, warmup=10, **kw)
    assert abi.last_kernel() == want
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=steps, calls=2, **kw)
    assert abi.last_kernel() == want


@pytest.mark.parametrize("n,p,intercept", [(10241, 3, True), (20000, 3, True), (12001, 3, False), (30000, 2, True), (5121, 5, True), (6500, 6, True), (9000, 4, False), (6000, 7, True), (8200, 7, False), (7000, 3, True), (10000, 3, True), (3100, 5, True), (2000, 3, True), (700, 1, False), (1000, 6, True)])
def test_adaptive_kernels_on_the_streamed_mfma_evaluation(E, O, monkeypatch, n, p, intercept):
    """mh_sweep_mfma_ad (round 4): kernel_adapt / kernel_ram beyond mh_sweep_spec's registers -- the streamed MFMA evaluation of
    all four chains of a workgroup, then one step of the register-row adaptive owners between barriers.  k = 5 (the
    compile-time owner) and the generic k <= 8 owner, with and without intercept, ragged last slot, continuation calls."""
    set_knob(monkeypatch, "shard", "0")     # (few chains on long data would take the long-data form: test_few_chains_on_long_data)
    from fmcmc_amd import _abi as abi
    X, y = synth_linreg(n, p, 6100 + n + p)
    k = p + 1 + (1 if intercept else 0)
    init = jitter_init(([0.0] if intercept else []) + [0.0] * p + [float(np.std(y))], 6, 97 + p)
    init[:, -1] = np.abs(init[:, -1])
    unbounded = "spec" if n <= (10240 if p <= 3 else (5120 if p <= 5 else 4096)) else "mfma-adaptive"   # (the bounded kernel_ram comes here from any n)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, k, init, nsteps=80, calls=2, warmup=20, intercept=intercept)
    assert abi.last_kernel() == unbounded
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=60, calls=2, intercept=intercept)
    assert abi.last_kernel() == unbounded
    # kernel_ram with bounds (the advanced-features vignette's kernel_ram(lb = c(NA, NA, NA, .001))): the decision takes f of the
    # REFLECTED proposal -- a second evaluation of the workgroup's four chains in the steps in which a reflection moved one
    rb, ro = run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=70, calls=2, intercept=intercept,
                      lb=[-0.25] * (k - 1) + [0.3], ub=[0.25] * (k - 1) + [float(np.std(y)) + 0.2])
    assert abi.last_kernel() == unbounded and ro.accept_count.sum() > 0      # (round 5: within mh_sweep_spec's registers, there)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=40, intercept=intercept, lb=[-50.0] * (k - 1) + [0.001], ub=50.0)   # (never reflects)
    assert abi.last_kernel() == unbounded
    set_knob(monkeypatch, "specbnd", "0")
    rb, ro = run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=70, calls=2, intercept=intercept,
                      lb=[-0.25] * (k - 1) + [0.3], ub=[0.25] * (k - 1) + [float(np.std(y)) + 0.2])
    assert abi.last_kernel() == "mfma-adaptive" and ro.accept_count.sum() > 0
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=40, intercept=intercept, lb=[-50.0] * (k - 1) + [0.001], ub=50.0)   # (never reflects)
    assert abi.last_kernel() == "mfma-adaptive"


@pytest.mark.parametrize("chains,n,p,intercept", [(1, 100, 3, True), (5, 100, 3, True), (300, 1000, 2, True), (700, 3000, 3, False),
                                                  (1001, 777, 5, True), (1030, 100, 6, False), (6, 4000, 7, False), (260, 3500, 7, True)])
def test_bounded_kernel_ram_on_the_wave_specialised_kernel_and_its_latency_forms(E, O, monkeypatch, chains, n, p, intercept):
    """Round 5: kernel_ram with bounds (R/kernel_ram.R:123-157 + R/mcmc.R:749-753; the advanced-features vignette's
    kernel_ram(lb = c(NA, NA, NA, .001))) on mh_sweep_spec's register owners -- two evaluation slots per step, the second one taken only
    when the reflection moved the proposal (SpecSyncB).  Tight bounds (most steps reflect), the vignette's kind of bound (hardly ever), one
    to four chains per workgroup, a ragged last workgroup, two calls with burn-in and thinning, step windows: the oracle's bits."""
    from fmcmc_amd import _abi as abi
    X, y = synth_linreg(n, p, 7300 + n + p)
    k = p + 1 + (1 if intercept else 0)
    init = jitter_init(([0.0] if intercept else []) + [0.0] * p + [float(np.std(y))], chains, 41 + p)
    init[:, -1] = np.abs(init[:, -1])
    steps = 90 if chains < 100 else 30
    want = "spec" if chains > 768 else "spec-lat%d" % ((chains + 255) // 256)
    tight = dict(lb=[-0.25] * (k - 1) + [0.3], ub=[0.25] * (k - 1) + [float(np.std(y)) + 0.2])
    init_t = np.clip(init, np.array(tight["lb"]) + 0.01, np.array(tight["ub"]) - 0.01)
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init_t, nsteps=steps, calls=2, burnin=3, thin=2, intercept=intercept, **tight)
    assert abi.last_kernel() == want, abi.last_kernel()
    assert ro.accept_count.sum() > 0
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=steps, intercept=intercept, lb=[-50.0] * (k - 1) + [0.001], ub=50.0, freq=2, warmup=3)
    assert abi.last_kernel() == want
    set_knob(monkeypatch, "window", "16")
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init_t, nsteps=steps, calls=2, intercept=intercept, **tight)
    assert abi.last_kernel() == want


@pytest.mark.parametrize("n,p,intercept,fix", [(3000, 9, True, False), (10000, 11, True, False), (2049, 12, False, False), (1537, 13, True, True),
                                               (6000, 14, True, False), (12000, 3, True, True), (6000, 6, True, True), (20000, 8, False, True),
                                               (900, 9, True, False), (1500, 13, True, False), (513, 8, False, True)])
def test_adaptive_kernels_with_their_matrices_in_lds_on_the_streamed_mfma_evaluation(E, O, monkeypatch, n, p, intercept, fix):
    """mh_sweep_mfma_ad<KIND, NG, -1> (round 4): kernel_adapt / kernel_ram with 8 .. 15 covariates (k <= 16), or with a fixed
    parameter beyond the wave-specialised kernel's range -- the owners that keep their matrices in LDS (spec_owner_adaptive, any
    k <= 16) between the barriers of the streamed MFMA evaluation, three / four operand groups per slot.  These calls ran on the
    general kernel (tools/dispatch_audit.py: 22 .. 55 us per step at n = 1e4)."""
    set_knob(monkeypatch, "shard", "0")     # (few chains on long data would take the long-data form: test_few_chains_on_long_data)
    from fmcmc_amd import _abi as abi
    # (round 5: 8 .. 14 covariates on up to 2048 observations run on the wave-specialised kernel -- register owners of width k, or, with
    #  a fixed parameter, the same owners in LDS there; knob specwide=0 keeps this test on the kernel it was written for)
    want = "mfma-adaptive"
    set_knob(monkeypatch, "specwide", "0")
    X, y = synth_linreg(n, p, 6400 + n + p, beta=np.linspace(0.7, -0.7, p + 1))
    k = p + 1 + (1 if intercept else 0)
    init = jitter_init(([0.0] if intercept else []) + [0.0] * p + [float(np.std(y))], 6, 98 + p)
    init[:, -1] = np.abs(init[:, -1])
    fixed = [False] * k
    if fix:
        fixed[1] = True
    run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, k, init, nsteps=70, calls=2, warmup=15, intercept=intercept, fixed=fixed)
    assert abi.last_kernel() == want
    run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, k, init, nsteps=50, warmup=10, intercept=intercept, fixed=fixed, lb=-4.0, ub=9.0, until=30.0)
    assert abi.last_kernel() == want
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=60, calls=2, intercept=intercept, fixed=fixed, burnin=2, thin=3)
    assert abi.last_kernel() == want
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=40, intercept=intercept, fixed=fixed, ram_qfun=1, warmup=4, freq=2)
    assert abi.last_kernel() == want


@pytest.mark.parametrize("k", [65, 100, 128])
@pytest.mark.parametrize("kind_name", ["ram", "adapt", "normal", "ram_bounded"])
def test_more_parameters_than_a_wavefront_has_lanes(E, O, k, kind_name):
    """64 < k <= 128 (round 4: FMCMC_MAX_K was 64; R/kernel_ram.R:93-121 and R/kernel_adapt.R:87-115 take any k, the authors'
    benchmark is k = 100): mh_sweep_bigk, one workgroup per chain, packed triangular matrices -- the oracle's bits for
    kernel_ram (product-form update, prefix sums over 128 rows), kernel_adapt (recursive covariance + Cholesky),
    kernel_normal_reflective, bounded kernel_ram (second evaluation), two consecutive calls, a fixed parameter."""
    from fmcmc_amd import _abi as abi
    p = k - 2
    n = 700 + k
    X, y = synth_linreg(n, p, 8000 + k, beta=np.linspace(1.0, -1.0, p + 1), sigma=2.0)
    init = jitter_init(list(np.linspace(1.0, -1.0, p + 1)) + [2.0], 3, 600 + k)
    init[:, -1] = np.abs(init[:, -1])
    fixed = [False] * k
    fixed[3] = (k == 100)                       # one case with a fixed parameter (kf = k - 1)
    if kind_name == "ram":
        run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=40, calls=2, fixed=fixed)
    elif kind_name == "ram_bounded":
        run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=30, calls=2, lb=-1.5, ub=2.5, fixed=fixed)
    elif kind_name == "adapt":
        run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, k, init, nsteps=40, calls=2, warmup=10, fixed=fixed)
    else:
        run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL_REFLECTIVE, k, init, nsteps=50, burnin=4, thin=3, calls=2, scale=0.01,
                 lb=-3.0, ub=4.0, fixed=fixed)
    assert abi.last_kernel() == "big-k"


def test_big_k_kernel_with_the_logistic_family(E, O):
    """k = 70 logistic regression (69 covariates + intercept) on mh_sweep_bigk: the all-family evaluation with the g table read
    from global memory, the scaled coefficient copies and the data-only sums, normal and RAM kernels."""
    from fmcmc_amd import _abi as abi
    rng = np.random.default_rng(70)
    n, p = 900, 69
    X = rng.standard_normal((n, p)) * 0.3
    beta = rng.uniform(-0.5, 0.5, p + 1)
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-(beta[0] + X @ beta[1:])))).astype(np.float64)
    init = jitter_init(beta, 3, 71)
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL, p + 1, init, nsteps=40, calls=2, prior_div=8.0, scale=0.01)
    assert abi.last_kernel() == "big-k"
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_RAM, p + 1, init, nsteps=30, prior_div=8.0)
    assert abi.last_kernel() == "big-k"


def test_what_the_big_k_kernel_refuses(E, O):
    """Above 64 parameters the single-parameter schemes, the mirror kernels and the windowed / strided kernel_adapt are
    refused with a message (FMCMC_ERR_UNSUPPORTED), not mis-run."""
    from fmcmc_amd import _abi as abi
    k = 70
    X, y = synth_linreg(500, k - 2, 5, beta=np.linspace(1.0, -1.0, k - 1))
    gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
    big = E.DBL_MAX
    z, o = np.zeros(k), np.ones(k)
    for kw in (dict(kind=abi.KERNEL_NORMAL, scheme=abi.SCHEME_ORDERED), dict(kind=abi.KERNEL_NMIRROR), dict(kind=abi.KERNEL_ADAPT, bw=20, warmup=30)):
        kind = kw.pop("kind")
        gk = E.KernelSpec(kind, k, z, 0.01 * o, -big * o, big * o, np.zeros(k, np.uint8), **kw)
        with pytest.raises(ValueError, match="k = 70 > 64 parameters"):
            E.sweep(gm, gk, E.ChainState(np.tile(np.r_[np.zeros(k - 1), 2.0], (2, 1)), k), 20)


def test_small_shape_mfma_equals_wave_specialised_kernel(E, O, monkeypatch):
    """README-size data (p = 1, n ~ 1000): the MFMA kernel (default) and the wave-specialised VALU kernel it replaced there
    (knob mfma=0) give the oracle's bits, normal and reflective kernels."""
    X, y = synth_linreg(1000, 1, 4242)
    init = jitter_init([0.0, 0.0, float(np.std(y))], 7, 77)
    init[:, -1] = np.abs(init[:, -1])
    out = []
    for mf in ("1", "0"):
        set_knob(monkeypatch, "mfma", mf)
        a, _ = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 3, init, nsteps=200, burnin=10, thin=3, calls=2, scale=0.05)
        b, _ = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL_REFLECTIVE, 3, init, nsteps=150, scale=0.5, lb=[-4, -4, 0.5], ub=[6, 6, 8.0])
        out.append((a.samples.cpu().numpy(), b.samples.cpu().numpy()))
    assert _bits_equal(out[0][0], out[1][0]) and _bits_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("form", ["chain-sharded", "observation-sharded"])
@pytest.mark.parametrize("kind_name", ["adapt", "adapt_window", "ram", "ram_bounded", "ram_constr"])
def test_logistic_adaptive_kernels_on_the_logistic_kernels(E, O, monkeypatch, kind_name, form):
    """kernel_adapt / kernel_ram on a logistic model (the workflow vignette's own combination) run on the logistic-only
    instantiations since round 4 -- g table in LDS, both the chain-sharded loop and the observation-sharded evaluation (the
    bounded kernel_ram on the chain-sharded one only) -- with 6 and 520 chains, two calls: the oracle's bits."""
    from fmcmc_amd import _abi as abi
    set_knob(monkeypatch, "shard", "1" if form == "observation-sharded" else "0")
    rng = np.random.default_rng(177)
    n, p = 2100, 4
    X = rng.standard_normal((n, p))
    beta = np.array([-0.5, 1.0, -1.0, 0.5, 0.25])
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-(beta[0] + X @ beta[1:])))).astype(np.float64)
    for chains, steps in ((6, 70), (520, 16)):
        init = jitter_init(beta, chains, 19)
        kw = dict(nsteps=steps, calls=2, prior_div=8.0)
        if kind_name == "adapt":
            run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_ADAPT, 5, init, warmup=8, **kw)
        elif kind_name == "adapt_window":
            run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_ADAPT, 5, init, warmup=8, bw=6, Sd=0.7, freq=2, lb=-3.0, ub=3.0, **kw)
        elif kind_name == "ram":
            run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_RAM, 5, init, **kw)
        elif kind_name == "ram_bounded":
            run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_RAM, 5, init, lb=-1.2, ub=1.2, **kw)
        else:
            run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_RAM, 5, init, freq=2,
                     constr=(np.abs(np.subtract.outer(np.arange(5), np.arange(5))) <= 1).astype(float), **kw)
        # (the bounded kernel_ram evaluates a second time only where a proposal was reflected: chain-sharded form only)
        # (round 5: plain kernel_adapt / kernel_ram with up to eight parameters run the sweep with the register owner, mh_sweep_logit2a)
        sharded = "logistic-shadow" if kind_name in ("adapt", "ram") else "logistic-sharded"
        assert abi.last_kernel() == (sharded if form == "observation-sharded" and kind_name != "ram_bounded" else "streamed-logistic")


@pytest.mark.parametrize("kind_name", ["adapt", "ram", "normal_ordered"])
def test_logistic_on_the_general_kernel(E, O, monkeypatch, kind_name):
    """Logistic model on the all-family streamed kernel (knob streamed=1; until round 4 the adaptive kernels ran there): its g
    table is read from global memory instead of LDS -- same bits as the oracle."""
    from fmcmc_amd import _abi as abi
    set_knob(monkeypatch, "streamed", "1")
    rng = np.random.default_rng(77)
    n, p = 1300, 3
    X = rng.standard_normal((n, p))
    beta = np.array([-0.5, 1.0, -1.0, 0.5])
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-(beta[0] + X @ beta[1:])))).astype(np.float64)
    init = jitter_init(beta, 5, 9)
    if kind_name == "adapt":
        run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_ADAPT, 4, init, nsteps=120, calls=2, warmup=30, prior_div=8.0)
    elif kind_name == "ram":
        run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_RAM, 4, init, nsteps=120, calls=2, prior_div=8.0)
    else:
        run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL, 4, init, nsteps=150, prior_div=8.0, scale=0.1, scheme="ordered")
    assert abi.last_kernel() == "streamed"


SHARDED = ("streamed-wide-sharded-mfma", "wide-dataflow")     # the sequential and the dataflow form of the observation-sharded sweep


@pytest.fixture(params=["dataflow", "dataflow4", "sequential"])
def wide_form(request, monkeypatch):
    """The forms of the observation-sharded sweep (mh_wide2.hpp with two or -- knob groups=4, slower, kept as a measured
    alternative -- four chain groups / eval_sharded): knob wide2=0 keeps the sequential one."""
    set_knob(monkeypatch, "wide2", "0" if request.param == "sequential" else "1")
    if request.param == "dataflow4":
        set_knob(monkeypatch, "groups", "4")
    return "dataflow" if request.param == "dataflow4" else request.param


@pytest.mark.parametrize("chains", [2, 3, 37, 256])
def test_dataflow_form_with_few_chains(E, O, chains):
    """kernel_ram on a wide model with one chain per CU or fewer (round 5): the dispatcher itself -- no knob -- puts two chains on a
    workgroup so that the sweep runs in the dataflow form (half of the workgroups, or nearly all of them, hold no chain and
    evaluate like the others): C4's shape at 256 chains 20.6 -> 15.0 us per step.  The oracle's bits, two calls."""
    import torch
    from fmcmc_amd import _abi as abi
    X, y = synth_linreg(6000, 20, 4747, beta=np.linspace(1.0, -1.0, 21))
    init = jitter_init(list(np.linspace(1.0, -1.0, 21)) + [4.0], chains, 8)
    init[:, -1] = np.abs(init[:, -1])
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, 22, init, nsteps=26, calls=2, burnin=3, thin=2)
    if torch.cuda.get_device_properties(0).multi_processor_count >= 256:
        assert abi.last_kernel() == "wide-dataflow"


def test_ram_families_in_the_dataflow_form(E, O, monkeypatch):
    """kernel_ram's qfun / eta families through mh_sweep_wide2 (its owners draw the variates themselves)."""
    import torch
    from fmcmc_amd import _abi as abi
    set_knob(monkeypatch, "cw", "2"); set_knob(monkeypatch, "shard", "1"); set_knob(monkeypatch, "wide2", "1")
    X, y = synth_linreg(1500, 20, 4242, beta=np.linspace(1.0, -1.0, 21))
    init = jitter_init(list(np.linspace(1.0, -1.0, 21)) + [4.0], 512, 6)
    init[:, -1] = np.abs(init[:, -1])
    for fam in (dict(ram_qfun=1, ram_eta_exp=0.8), dict(ram_qfun=2, ram_df=5.0)):
        run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, 22, init, nsteps=20, calls=2, **fam)
        if torch.cuda.get_device_properties(0).multi_processor_count >= 256:
            assert abi.last_kernel() == "wide-dataflow"


@pytest.mark.parametrize("n,p,chains,intercept", [(10241, 20, 6, True), (12289, 33, 40, True), (16385, 17, 5, False), (20480, 48, 9, True),
                                                  (20481, 16, 300, True), (24576, 60, 7, True), (23000, 21, 513, True)])
def test_observation_sharded_slices_beyond_forty_observations(E, O, monkeypatch, n, p, chains, intercept):
    """Wide linear models with 10,240 < n <= 24,576 (round 4): the slice of a workgroup -- 41 .. 96 observations -- as four to six
    M-tiles of the matrix-core slice product (it was three: tools/dispatch_audit.py found p = 48, n = 2e4 on the chain-sharded
    kernel at 6.4x the time per step of n = 1e4).  The tile-count edges, ragged last slots, few and many chains (consecutive
    launches above 512), kernel_normal, kernel_normal_reflective and kernel_ram (sequential form), two calls: the oracle's bits."""
    import torch
    from fmcmc_amd import _abi as abi
    set_knob(monkeypatch, "shard", "1")
    rng = np.random.default_rng(n + p)
    beta = rng.uniform(-1.0, 1.0, p + 1)
    X = rng.standard_normal((n, p))
    y = (beta[0] if intercept else 0.0) + X @ beta[1:] + 2.0 * rng.standard_normal(n)
    k = p + 1 + (1 if intercept else 0)
    init = jitter_init(list(beta[(0 if intercept else 1):]) + [2.0], chains, 5 + p)
    init[:, -1] = np.abs(init[:, -1]) + 0.1
    steps = int(max(6, min(30, 1.2e8 / (chains * n * p * 2))))
    full = torch.cuda.get_device_properties(0).multi_processor_count >= 256
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, k, init, nsteps=steps, calls=2, scale=0.004, intercept=intercept)
    if full:
        assert abi.last_kernel() == "streamed-wide-sharded-mfma"
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL_REFLECTIVE, k, init, nsteps=steps, burnin=1, thin=2, scale=0.05, lb=-1.2, ub=2.6, intercept=intercept)
    if full:
        assert abi.last_kernel() == "streamed-wide-sharded-mfma"
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, k, init, nsteps=steps, calls=2, intercept=intercept)
    if full:
        assert abi.last_kernel() == "streamed-wide-sharded-mfma"


@pytest.mark.parametrize("n,p,chains,intercept", [(4096, 3, 1, True), (5001, 1, 3, True), (30000, 3, 4, True), (65537, 2, 2, False), (100000, 3, 4, True),
                                                  (200001, 5, 1, True), (50000, 8, 5, True), (40000, 14, 2, True), (300000, 0, 3, True), (20000, 3, 40, True),
                                                  (24577, 16, 3, True), (30000, 48, 2, True), (60001, 33, 1, False), (25000, 62, 4, True)])
def test_few_chains_on_long_data(E, O, monkeypatch, n, p, chains, intercept):
    """The long-data form of the observation-sharded evaluation (shard_long, round 4): up to 64 chains, every one of the 256
    workgroups evaluates its 1/256 of the observations for all of them -- residuals into LDS by all threads, then one thread per
    (chain, canonical lane) walks its slots in order -- instead of one workgroup walking the whole data set.  Forced by knob
    shard=1 at the small sizes; ragged n, no intercept, p = 0 (iid Normal) .. 14 and wide models (p = 16 .. 62) beyond the matrix-core
    slices' 24,576 observations, chain groups (40 chains at n = 20,000 need two LDS
    passes... or one), kernel_normal, kernel_normal_reflective with an ordered scheme, kernel_ram, two calls: the oracle's bits."""
    import torch
    from fmcmc_amd import _abi as abi
    if torch.cuda.get_device_properties(0).multi_processor_count < 256:
        pytest.skip("the sharded forms need all 256 CUs")
    set_knob(monkeypatch, "shard", "1")
    rng = np.random.default_rng(n + 7 * p)
    beta = rng.uniform(-1.0, 1.0, p + 1)
    X = rng.standard_normal((n, p)) if p else None
    y = (beta[0] if intercept else 0.0) + (X @ beta[1:] if p else 0.0) + 2.0 * rng.standard_normal(n)
    k = p + 1 + (1 if intercept else 0)
    init = jitter_init(list(beta[(0 if intercept else 1):]) + [2.0], chains, 9 + p)
    init[:, -1] = np.abs(init[:, -1]) + 0.1
    steps = int(max(8, min(40, 1.0e8 / (chains * n * max(p, 1) * 2))))
    fam = O.FAM_LINREG if p else O.FAM_IID_NORMAL
    kw = dict(intercept=intercept) if p else {}
    run_both(E, O, fam, X, y, O.K_NORMAL, k, init, nsteps=steps, calls=2, scale=0.004, burnin=2, thin=2, **kw)
    assert abi.last_kernel() == "long-sharded"
    run_both(E, O, fam, X, y, O.K_NORMAL_REFLECTIVE, k, init, nsteps=steps, scale=0.05, lb=-1.2, ub=2.6, scheme="ordered", **kw)
    assert abi.last_kernel() == "long-sharded"
    run_both(E, O, fam, X, y, O.K_RAM, k, init, nsteps=steps, calls=2, **kw)
    assert abi.last_kernel() == "long-sharded"
    run_both(E, O, fam, X, y, O.K_ADAPT, k, init, nsteps=steps, calls=2, warmup=3, lb=[-40.0] * (k - 1) + [0.05], ub=40.0, **kw)
    assert abi.last_kernel() == "long-sharded"


@pytest.mark.parametrize("n,p,chains,intercept", [(4096, 3, 1, True), (9001, 1, 3, False), (30000, 5, 4, True), (65537, 8, 2, True), (100000, 5, 4, True),
                                                  (200001, 2, 1, True), (40000, 12, 3, True), (25000, 16, 2, False), (20000, 5, 40, True)])
def test_few_chains_on_long_data_logistic(E, O, monkeypatch, n, p, chains, intercept):
    """The long-data form for the logistic family (shard_long<LOGISTIC>): the terms g(|eta|) of the slice into LDS by all threads
    (the checked form: one column scaled by 12 puts |eta| beyond the table), then one thread per (chain, canonical lane) adds them in
    slot order.  1 .. 40 chains (LDS groups), p up to 16, ragged n, all four proposal families, two calls: the oracle's bits."""
    import torch
    from fmcmc_amd import _abi as abi
    if torch.cuda.get_device_properties(0).multi_processor_count < 256:
        pytest.skip("the sharded forms need all 256 CUs")
    set_knob(monkeypatch, "shard", "1")
    rng = np.random.default_rng(3 * n + p)
    X = rng.standard_normal((n, p))
    if p == 8:
        X[:, 0] *= 12.0
    k = p + (1 if intercept else 0)
    beta = rng.uniform(-1.0, 1.0, k)
    eta = (beta[0] if intercept else 0.0) + X @ beta[(1 if intercept else 0):]
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-eta))).astype(np.float64)
    init = jitter_init(beta, chains, 50 + p)
    steps = int(max(8, min(40, 6.0e7 / (chains * n * max(p, 1) * 2))))
    kw = dict(intercept=intercept, prior_div=8.0)
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL, k, init, nsteps=steps, calls=2, scale=0.01, burnin=2, thin=2, **kw)
    assert abi.last_kernel() == "long-sharded"
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_NORMAL_REFLECTIVE, k, init, nsteps=steps, scale=0.1, lb=-1.5, ub=1.5, **kw)
    assert abi.last_kernel() == "long-sharded"
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_ADAPT, k, init, nsteps=steps, calls=2, warmup=3, **kw)
    assert abi.last_kernel() == "long-sharded"
    run_both(E, O, O.FAM_LOGISTIC, X, y, O.K_RAM, k, init, nsteps=steps, **kw)
    assert abi.last_kernel() == "long-sharded"


@pytest.mark.parametrize("form", ["logistic-sharded", "wide-sequential", "wide-dataflow", "long-sharded"])
def test_a_lost_hand_over_ends_in_status_5_not_in_a_hang(E, monkeypatch, form):
    """The grid-wide hand-overs of the observation-sharded kernels, with a FAULT: knob mode=512 makes workgroup 1 skip ONE
    arrival (epoch 3) and shortens every spin bound.  Every waiter must give up, every loop must run out and the grid must
    drain: the call returns, chains carry status 5 (FMCMC_CHAIN_SYNC_TIMEOUT) and the host raises 'results invalid' instead
    of handing out samples -- the path that had only ever been argued, not run."""
    import time
    import torch
    from fmcmc_amd import _abi as abi
    if torch.cuda.get_device_properties(0).multi_processor_count < 256:
        pytest.skip("the sharded forms need all 256 CUs")
    set_knob(monkeypatch, "shard", "1")
    set_knob(monkeypatch, "mode", "512")
    big = E.DBL_MAX
    if form == "logistic-sharded":
        rng = np.random.default_rng(3)
        n, p, C = 3000, 3, 40
        X = rng.standard_normal((n, p)); y = (rng.uniform(size=n) < 0.4).astype(np.float64)
        gm = E.DeviceModel(abi.FAM_LOGISTIC, X, y, intercept=True, guard=False, prior_div=8.0)
        k = p + 1
        gk = E.KernelSpec(abi.KERNEL_NORMAL, k, np.zeros(k), np.full(k, 0.05), np.full(k, -big), np.full(k, big), np.zeros(k, np.uint8))
        init = 0.1 * rng.standard_normal((C, k))
    elif form == "long-sharded":
        X, y = synth_linreg(30000, 3, 4243)
        k, C = 5, 4
        gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
        gk = E.KernelSpec(abi.KERNEL_NORMAL, k, np.zeros(k), np.full(k, 0.02), np.full(k, -big), np.full(k, big), np.zeros(k, np.uint8))
        init = jitter_init([0.0] * 4 + [float(np.std(y))], C, 6)
        init[:, -1] = np.abs(init[:, -1])
    else:
        set_knob(monkeypatch, "cw", "2")
        set_knob(monkeypatch, "wide2", "1" if form == "wide-dataflow" else "0")
        X, y = synth_linreg(1500, 20, 4242, beta=np.linspace(1.0, -1.0, 21))
        k, C = 22, 512
        gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
        gk = E.KernelSpec(abi.KERNEL_RAM, k, np.zeros(k), np.ones(k), np.full(k, -big), np.full(k, big), np.zeros(k, np.uint8))
        init = jitter_init(list(np.linspace(1.0, -1.0, 21)) + [4.0], C, 6)
        init[:, -1] = np.abs(init[:, -1])
    st = E.ChainState(init, gk.kf)
    t0 = time.time()
    r = E.sweep(gm, gk, st, 40, seed=7, check=False)
    torch.cuda.synchronize()
    assert time.time() - t0 < 120.0
    want = {"logistic-sharded": "logistic-shadow", "wide-sequential": "streamed-wide-sharded-mfma", "wide-dataflow": "wide-dataflow",
            "long-sharded": "long-sharded"}[form]
    assert abi.last_kernel() == want
    status = r.status.cpu().numpy()
    assert (status == abi.CHAIN_SYNC_TIMEOUT).any() and set(np.unique(status)) <= {0, abi.CHAIN_SYNC_TIMEOUT}
    with pytest.raises(RuntimeError, match="results of this call are invalid"):
        E.raise_on_chain_error(r)
    # the device is fine afterwards: the same call without the fault runs clean
    set_knob(monkeypatch, "mode", "0")
    r2 = E.sweep(gm, gk, E.ChainState(init, gk.kf), 40, seed=7, check=False)
    torch.cuda.synchronize()
    assert int(r2.status.abs().sum().item()) == 0


@pytest.mark.parametrize("chains,cw,n,p,intercept", [
    (256, "1", 10000, 16, True),      # 256 workgroups x 2 canonical lanes, 20 slots: the full slice of 40 observations
    (128, "1", 4099, 21, True),       # 128 workgroups x 4 lanes, ragged last slot
    (512, "2", 1023, 33, True),       # two chains per workgroup, n < 2 x 512
    (255, "2", 2500, 49, False),      # last workgroup holds one chain; widest slice (49 columns); no intercept
    (300, "2", 7000, 17, True),       # 150 workgroups with chains + 106 without: every chain count is eligible
    (77, "1", 10240, 23, True),       # 77 of 256 workgroups hold a chain
    (130, "1", 9000, 62, True),       # k = 64, the widest model: the slice (19.8 KB) no longer fits the scalar cache
])
def test_observation_sharded_evaluation(E, O, monkeypatch, wide_form, chains, cw, n, p, intercept):
    """Wide linear models whose workgroups split the 512 canonical lanes evenly evaluate observation-sharded
    (eval_sharded, cooperative launch, two grid barriers per step): the oracle's bits for the normal, reflective and RAM
    kernels, continued over two calls, and the same bits as the chain-sharded kernel (knob shard=0)."""
    import torch
    from fmcmc_amd import _abi as abi
    set_knob(monkeypatch, "cw", cw)
    set_knob(monkeypatch, "shard", "1")      # every eligible shape, also where the cost model prefers chain-sharded
    nb = p + (1 if intercept else 0)
    # co-residency needs one CU per workgroup of the launch (256, or exactly 128): a partitioned GPU falls back
    full = torch.cuda.get_device_properties(0).multi_processor_count >= 256
    # (the dataflow form needs two chains per workgroup and 256 workgroups; everything else runs the sequential form)
    sharded = ("wide-dataflow" if (wide_form == "dataflow" and cw == "2" and chains > 256) else "streamed-wide-sharded-mfma") if full else "streamed-wide"
    X, y = synth_linreg(n, p, 9100 + n + p, beta=np.linspace(1.0, -1.0, p + 1))
    init = jitter_init(list(np.linspace(1.0, -1.0, p + 1))[(0 if intercept else 1):] + [4.0], chains, n + p)
    init[:, -1] = np.abs(init[:, -1])
    kw = dict(intercept=intercept)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, nb + 1, init, nsteps=24, burnin=2, thin=2, calls=2, scale=0.01, **kw)
    assert abi.last_kernel() == sharded
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL_REFLECTIVE, nb + 1, init, nsteps=16, scale=0.2, lb=-3.0, ub=6.0, **kw)
    assert abi.last_kernel() == sharded
    a, _ = run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, nb + 1, init, nsteps=24, calls=2, **kw)
    assert abi.last_kernel() == sharded
    set_knob(monkeypatch, "shard", "0")
    b, _ = run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, nb + 1, init, nsteps=24, calls=2, **kw)
    assert abi.last_kernel() == "streamed-wide"
    assert _bits_equal(a.samples.cpu().numpy(), b.samples.cpu().numpy())


@pytest.mark.parametrize("chains,cw,n,p,intercept", [
    (512, "2", 10240, 48, True),      # 20 full slots: 10 values per lane group, every one valid; 12 full K-blocks
    (512, "2", 9300, 47, True),       # 19 slots: value 9 of the second half missing, slot 18 ragged
    (512, "2", 8193, 45, False),      # 17 slots (9 values per group, the last of the second half missing); 1 + 3 padded columns
    (512, "2", 9216, 46, True),       # 18 slots, all of them full (n = 18 x 512): 9 values per group, none ragged
    (128, "1", 5000, 46, True),       # sequential form with four lanes per workgroup (one lane group per canonical lane): 10 values
])
@pytest.mark.parametrize("form", ["dataflow", "sequential"])
def test_slice_product_with_the_half_padded_third_tile_as_4x4x4(E, O, monkeypatch, form, chains, cw, n, p, intercept):
    """shard_columns_mfma's form T10: three M-tiles of which the third carries 8 rows only, at the width
    with the compile-time K-block count (45..48 covariates) -- its two live D registers come from two v_mfma_f64_4x4x4 per
    K-block.  The oracle's bits, and the same bits with the knob t10=0 (a plain third 16x16x4 tile)."""
    import torch
    from fmcmc_amd import _abi as abi
    if form == "dataflow" and cw != "2":
        pytest.skip("the dataflow form needs two chains per workgroup")
    set_knob(monkeypatch, "cw", cw); set_knob(monkeypatch, "shard", "1"); set_knob(monkeypatch, "wide2", "1" if form == "dataflow" else "0")
    full = torch.cuda.get_device_properties(0).multi_processor_count >= 256
    nb = p + (1 if intercept else 0)
    X, y = synth_linreg(n, p, 777 + n + p, beta=np.linspace(1.0, -1.0, p + 1))
    init = jitter_init(list(np.linspace(1.0, -1.0, p + 1))[(0 if intercept else 1):] + [4.0], chains, n + p)
    init[:, -1] = np.abs(init[:, -1])
    out = {}
    for t10 in ("1", "0"):
        set_knob(monkeypatch, "t10", t10)
        a, _ = run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, nb + 1, init, nsteps=12, calls=2, intercept=intercept)
        if full:
            assert abi.last_kernel() == ("wide-dataflow" if form == "dataflow" else "streamed-wide-sharded-mfma")
        b, _ = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, nb + 1, init, nsteps=10, scale=0.01, intercept=intercept)
        out[t10] = (a.samples.cpu().numpy(), b.samples.cpu().numpy())
    assert _bits_equal(out["1"][0], out["0"][0]) and _bits_equal(out["1"][1], out["0"][1])


def test_observation_sharded_long_run_equals_chain_sharded(E, monkeypatch, wide_form):
    """Config C4's shape (512 chains, n = 10,000, k = 50, kernel_ram), 400 steps: 800 grid barriers without a stale read --
    the sharded and the chain-sharded kernels return identical bits for every output."""
    import torch
    from fmcmc_amd import _abi as abi
    n, p, chains = 10000, 48, 512
    X, y = synth_linreg(n, p, 515, beta=np.linspace(1.0, -1.0, p + 1))
    init = jitter_init(list(np.linspace(1.0, -1.0, p + 1)) + [4.0], chains, 99)
    init[:, -1] = np.abs(init[:, -1])
    k = p + 2
    gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
    outs = []
    for sh in ("1", "0"):
        set_knob(monkeypatch, "shard", sh)
        gk = E.KernelSpec(abi.KERNEL_RAM, k, np.zeros(k), np.ones(k), np.full(k, -E.DBL_MAX), np.full(k, E.DBL_MAX),
                          np.zeros(k, np.uint8))
        st = E.ChainState(init, k)
        r = E.sweep(gm, gk, st, 400, seed=77, check=True)
        torch.cuda.synchronize()
        if torch.cuda.get_device_properties(0).multi_processor_count >= 256:
            assert abi.last_kernel() == (("wide-dataflow" if wide_form == "dataflow" else "streamed-wide-sharded-mfma") if sh == "1" else "streamed-wide")
        outs.append([t.cpu().numpy() for t in (r.samples, r.logpost, r.draws, r.accept_count, st.Sigma, st.theta0)])
    for u, v in zip(*outs):
        assert _bits_equal(u, v) if u.dtype == np.float64 else np.array_equal(u, v)


# ---------------------------------------------------------------------------------------------------------------------
# Randomised sweep over the dispatcher: shapes, chain counts and options drawn from a fixed seed, so that combinations
# nobody wrote a dedicated test for (a kernel variant x a ragged shape x a fixed mask x thinning x a continued call)
# still have to give the oracle's bits.  Every case is sized so that the scalar oracle needs well under a second.
# ---------------------------------------------------------------------------------------------------------------------
def _random_case(seed):
    rng = np.random.default_rng(900000 + seed)
    fam = rng.choice(["linreg", "linreg", "linreg", "logistic", "iid"])
    n = int(rng.choice([1, 7, 511, 512, 513, 1000, 1024, 2049, 5000, 9999, 10240, 10241]))
    p = int(rng.choice([1, 1, 2, 3, 3, 4, 7, 8, 15, 16, 17, 24, 31]))
    intercept = bool(rng.integers(0, 2))
    chains = int(rng.choice([1, 3, 4, 5, 8, 31, 64, 128, 129, 256]))
    if fam == "iid":
        p, intercept = 0, True
    if fam == "logistic":
        p = min(p, 9)
    kind = str(rng.choice(["normal", "reflective", "adapt", "ram", "unif", "unif_reflective", "nmirror", "umirror"]))
    if fam == "linreg":
        k = p + int(intercept) + 1
    elif fam == "logistic":
        k = p + int(intercept)
    else:
        k = 2
    # keep the oracle's work bounded: chains x n x max(p, 1) x steps <= ~1.5e8 multiply-adds
    nsteps = int(max(6, min(90, 1.5e8 / (chains * n * max(p, 1) * 2))))
    burnin = int(rng.integers(0, max(1, nsteps // 3)))
    thin = int(rng.choice([1, 1, 2, 3]))
    if burnin + thin > nsteps - 1:
        burnin, thin = 0, 1
    fixed = np.zeros(k, bool)
    if k > 2 and rng.random() < 0.35:
        fixed[rng.choice(k - 1, size=int(rng.integers(1, max(2, (k - 1) // 2))), replace=False)] = True
    scheme = "joint"
    if kind in ("normal", "reflective", "unif", "unif_reflective", "nmirror", "umirror") and rng.random() < 0.3:
        scheme = str(rng.choice(["ordered", "random"]))
    return dict(fam=fam, n=n, p=p, intercept=intercept, chains=chains, kind=kind, k=k, nsteps=nsteps, burnin=burnin,
                thin=thin, fixed=fixed, scheme=scheme, calls=int(rng.integers(1, 3)), seed=int(rng.integers(1, 10**6)),
                chain_base=int(rng.choice([0, 0, 5, 1000])))


@pytest.mark.parametrize("case", range(int(os.environ.get("FMCMC_TEST_RANDOM_CASES", "160"))))   # soak: 3000 passed (round 5, last code: 3000)
def test_randomised_dispatch_cases(E, O, case):
    c = _random_case(case)
    rng = np.random.default_rng(17 + case)
    n, p, k = c["n"], c["p"], c["k"]
    if c["fam"] == "linreg":
        beta = rng.uniform(-1.5, 1.5, p + 1)
        X = rng.standard_normal((n, p))
        y = (beta[0] if c["intercept"] else 0.0) + X @ beta[1:] + 2.0 * rng.standard_normal(n)
        base = list(beta[(0 if c["intercept"] else 1):]) + [2.0]
        fam, kw = O.FAM_LINREG, dict(intercept=c["intercept"])
        lb = [-40.0] * (k - 1) + [0.05]
    elif c["fam"] == "logistic":
        beta = rng.uniform(-1.0, 1.0, p + 1)
        X = rng.standard_normal((n, p))
        eta = (beta[0] if c["intercept"] else 0.0) + X @ beta[1:]
        y = (rng.uniform(size=n) < 1 / (1 + np.exp(-eta))).astype(np.float64)
        base = list(beta[(0 if c["intercept"] else 1):])
        fam, kw = O.FAM_LOGISTIC, dict(intercept=c["intercept"], prior_div=8.0)
        lb = [-6.0] * k
    else:
        X, y = None, 1.5 + 2.0 * rng.standard_normal(n)
        base = [1.0, 2.0]
        fam, kw = O.FAM_IID_NORMAL, {}
        lb = [-40.0, 0.05]
    init = jitter_init(base, c["chains"], 3 + case)
    if c["fam"] != "logistic":
        init[:, -1] = np.abs(init[:, -1]) + 0.1
    kind = {"normal": O.K_NORMAL, "reflective": O.K_NORMAL_REFLECTIVE, "adapt": O.K_ADAPT, "ram": O.K_RAM,
            "unif": O.K_UNIF, "unif_reflective": O.K_UNIF_REFLECTIVE, "nmirror": O.K_NMIRROR,
            "umirror": O.K_UMIRROR}[c["kind"]]
    opts = dict(fixed=c["fixed"], scheme=c["scheme"])
    if c["kind"] in ("normal", "reflective"):
        opts["scale"] = 0.03
    if c["kind"] in ("unif", "unif_reflective"):
        opts.update(min_=-0.04, max_=0.05)
    if c["kind"] in ("reflective", "unif_reflective") or (c["kind"] in ("adapt", "ram") and case % 2):
        opts.update(lb=lb, ub=40.0)
    if c["kind"] == "adapt":
        opts["warmup"] = 5
        if case % 3 == 0:
            opts["freq"] = 2
        if case % 5 == 0:
            opts["until"] = c["nsteps"] // 2
    if c["kind"] == "ram" and case % 3 == 0:
        opts.update(warmup=3, freq=2)
    if c["kind"] in ("nmirror", "umirror"):
        opts.update(mu=base, scale=0.1, warmup=c["nsteps"] // 2, nadapt=4, lb=lb, ub=40.0)
    run_both(E, O, fam, X, y, kind, k, init, nsteps=c["nsteps"], burnin=c["burnin"], thin=c["thin"], seed=c["seed"],
             chain_base=c["chain_base"], calls=c["calls"], **kw, **opts)


def test_randomised_sharded_shapes(E, O, monkeypatch):
    """Random wide models around the eligibility limits of the observation-sharded evaluation (128 / 256 workgroups, slices
    of <= 40 observations and <= 49 columns): whichever kernel the dispatcher picks, the oracle's bits; most of them
    must actually have run sharded."""
    import torch
    from fmcmc_amd import _abi as abi
    set_knob(monkeypatch, "shard", "1")      # every eligible shape, also where the cost model prefers chain-sharded
    picked = []
    for case in range(18):
        rng = np.random.default_rng(4200 + case)
        chains = int(rng.choice([128, 200, 255, 256, 257, 300, 511, 512, 700]))
        p = int(rng.integers(16, 50))
        lanes = 4 if chains == 128 else 2
        nmax = 512 * (40 // lanes)
        n = int(rng.choice([nmax, nmax - 1, nmax + 1, nmax // 2 + 3, 777, 512, 513]))
        intercept = bool(rng.integers(0, 2))
        k = p + int(intercept) + 1
        kind = [O.K_NORMAL, O.K_NORMAL_REFLECTIVE, O.K_RAM][case % 3]
        nsteps = int(max(5, min(40, 2.0e8 / (chains * n * p * 2))))
        beta = rng.uniform(-1.0, 1.0, p + 1)
        X = rng.standard_normal((n, p))
        y = (beta[0] if intercept else 0.0) + X @ beta[1:] + 2.0 * rng.standard_normal(n)
        init = jitter_init(list(beta[(0 if intercept else 1):]) + [2.0], chains, case)
        init[:, -1] = np.abs(init[:, -1]) + 0.1
        fixed = np.zeros(k, bool)
        if case % 4 == 1:
            fixed[rng.choice(k - 1, size=3, replace=False)] = True
        opts = dict(fixed=fixed, intercept=intercept)
        if kind == O.K_NORMAL:
            opts["scale"] = 0.01
        if kind == O.K_NORMAL_REFLECTIVE:
            opts.update(scale=0.05, lb=[-3.0] * (k - 1) + [0.05], ub=6.0)
        run_both(E, O, O.FAM_LINREG, X, y, kind, k, init, nsteps=nsteps, burnin=case % 3, thin=1 + case % 2, seed=100 + case,
                 chain_base=17 * case, calls=1 + case % 2, **opts)
        picked.append(abi.last_kernel())
    assert all(name.startswith("streamed-wide") or name == "wide-dataflow" for name in picked), picked
    if torch.cuda.get_device_properties(0).multi_processor_count >= 256:
        assert sum(name in SHARDED for name in picked) >= 12, picked


def test_sharded_evaluation_with_failing_chains(E, O, monkeypatch):
    """Chains that raise "fun(par) is undefined" (sigma < 0 without the guard) inside an observation-sharded sweep: they
    stop, their workgroups keep taking part in every grid-wide hand-over, the other chains are unaffected (oracle's bits,
    status, step and theta of the failure), for kernel_normal and kernel_ram."""
    import torch
    from fmcmc_amd import _abi as abi
    set_knob(monkeypatch, "shard", "1")
    n, p, chains = 3000, 20, 256
    rng = np.random.default_rng(77)
    beta = rng.uniform(-1.0, 1.0, p + 1)
    X = rng.standard_normal((n, p))
    y = beta[0] + X @ beta[1:] + 2.0 * rng.standard_normal(n)
    init = jitter_init(list(beta) + [2.0], chains, 5)
    init[:, -1] = np.abs(init[:, -1]) + 0.1
    init[::7, -1] = 0.004                       # every 7th chain: sigma steps below zero within a few proposals
    scale = np.full(p + 2, 0.001); scale[-1] = 0.05
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, p + 2, init, nsteps=40, guard=False, scale=scale)
    if torch.cuda.get_device_properties(0).multi_processor_count >= 256:
        assert abi.last_kernel() in SHARDED
    assert (ro.status == 1).sum() >= 10 and (ro.status == 0).sum() >= 200
    assert np.array_equal(rg.status_step.cpu().numpy(), ro.status_step)
    with pytest.raises(RuntimeError, match="undefined"):
        E.raise_on_chain_error(rg)
    init[::7, -1] = 2e-5
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, p + 2, init, nsteps=30, guard=False)
    assert (ro.status != 0).any() and (ro.status == 0).any()
    assert np.array_equal(rg.status_step.cpu().numpy(), ro.status_step)


@pytest.mark.parametrize("path", ["mfma", "spec"])
@pytest.mark.parametrize("window", ["32", "96"])
def test_step_windows_do_not_change_a_bit(E, O, monkeypatch, path, window):
    """A long call of the stream-fed kernels runs as consecutive step windows with a bounded RNG stream (mh_engine.hip,
    launch_sweep).  With the window forced down to 32 / 96 steps a call of a few hundred steps is cut 4 - 12 times: outputs,
    accept bitmap, counts, state and error reports equal the oracle's (inside run_both) -- burn-in and thinning that do not
    divide the window, a second call continuing the first, the reflective and the uniform kernel, a fixed parameter, and a
    chain that fails with a NaN in a LATER window (its status step is the call's step, not the window's)."""
    from fmcmc_amd import _abi as abi
    set_knob(monkeypatch, "window", window)
    if path == "spec":
        set_knob(monkeypatch, "mfma", "0")
    X, y = synth_linreg(10000, 3, 20260102)
    init = jitter_init([0, 0, 0, 0, float(np.std(y))], 9, 31)
    init[:, -1] = np.abs(init[:, -1])
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, init, nsteps=397, burnin=41, thin=7, calls=2, scale=0.02)
    assert abi.last_kernel() == path
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, init, nsteps=130, scale=0.03, fixed=[False, False, True, False, False])
    run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL_REFLECTIVE, 5, init, nsteps=301, burnin=3, thin=2, scale=0.3,
             lb=[-5, -5, -5, -5, 0.1], ub=[5, 5, 5, 5, 5.0], guard=False)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_UNIF, 5, init, nsteps=200, thin=3, min_=-0.03, max_=0.04)
    assert abi.last_kernel() == path
    # proposals of sigma with scale 2 around ~4.5 step below zero every few dozen steps: without the guard eight of the
    # nine chains fail, at steps 45 .. 280 of the call (one survives)
    rg, ro = run_both(E, O, O.FAM_LINREG, X, y, O.K_NORMAL, 5, init, nsteps=400, guard=False, scale=[0.02, 0.02, 0.02, 0.02, 2.0])
    assert (ro.status == 1).sum() >= 5 and (ro.status == 0).any() and ro.status_step[ro.status == 1].max() > 2 * int(window) + 1


def test_a_call_longer_than_4_gib_of_samples_stays_on_the_headline_kernel(E, monkeypatch):
    """BASELINE configs[1] with nsteps = 120,000: 4.9 GB of samples (and as much again of draws) -- until round 3 such a call
    left mh_sweep_mfma for the general kernel at the 4 GiB limit of its 32-bit offsets and of its materialised RNG stream.
    Now: (a) it runs on "mfma"; (b) the size-independent properties of the 10^4-step test hold (rows follow the accept bits,
    counts are popcounts); (c) a 64-chain shard with the same chain ids equals the general kernel (knob streamed=1) bit for
    bit.  (The time per step against a 10^4-step call's: tests/test_gpu_perf_guard.py.)"""
    import torch
    from fmcmc_amd import _abi as abi
    import bench
    chains, iters, k = 1024, 120000, 5
    X, y, init = bench.Config("c2").workload(chains, 0)
    gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
    gk = E.KernelSpec(abi.KERNEL_NORMAL, k, np.zeros(k), np.full(k, 0.02), np.full(k, -E.DBL_MAX),
                      np.full(k, E.DBL_MAX), np.zeros(k, np.uint8))

    def timed(nsteps, lo=0, hi=chains, **kw):
        st = E.ChainState(init[lo:hi], k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        r = E.sweep(gm, gk, st, nsteps, seed=bench.CHAIN_SEED, chain_base=lo, check=True, **kw)
        e1.record()
        torch.cuda.synchronize()
        return r, st, e0.elapsed_time(e1) * 1e3 / nsteps

    full, st_full, _ = timed(iters)
    assert abi.last_kernel() == "mfma"                                             # (a)
    assert full.samples.numel() * 8 > (4 << 30)
    # (b) structure, chunked over the chains (the bitmap expands to one byte per row)
    s64, d64 = full.samples.view(torch.int64), full.draws.view(torch.int64)
    bits = full.accept_bits.view(torch.int32)
    r_idx = torch.arange(iters, device=bits.device)
    assert torch.equal(s64[:, :, 0], torch.as_tensor(init, device=bits.device).view(torch.int64))
    for lo in range(0, chains, 64):
        acc = ((bits[lo:lo + 64][:, (r_idx >> 5)] >> (r_idx & 31)) & 1).bool()
        assert not bool(acc[:, 0].any())
        expect = torch.where(acc[:, None, 1:], d64[lo:lo + 64, :, 1:], s64[lo:lo + 64, :, :-1])
        assert torch.equal(s64[lo:lo + 64, :, 1:], expect)
        assert torch.equal(acc.sum(dim=1), full.accept_count[lo:lo + 64].to(torch.int64))
        del acc, expect
    # (c) the general kernel on a shard
    set_knob(monkeypatch, "streamed", "1")
    part, st_part, _ = timed(iters, 448, 512)
    assert abi.last_kernel() == "streamed"
    for name in ("samples", "logpost", "draws", "accept_count", "accept_bits"):
        assert torch.equal(getattr(part, name), getattr(full, name)[448:512]), name
    assert torch.equal(st_part.theta0, st_full.theta0[448:512]) and torch.equal(st_part.f0, st_full.f0[448:512])
    del part
    # ((d), the time per step of the long call against a 10^4-step call, lives in tests/test_gpu_perf_guard.py)


@pytest.mark.parametrize("window", ["32", "96"])
@pytest.mark.parametrize("kind", ["adapt", "ram"])
@pytest.mark.parametrize("shape", ["spec", "spec-lds-owner", "mfma-adaptive", "spec-lat1"])
def test_step_windows_of_the_adaptive_kernels(E, O, monkeypatch, kind, window, shape):
    """Round 5: kernel_adapt / kernel_ram in step windows (they used to materialise the whole stream and leave for the general
    kernel at 8 GiB).  R/kernel_adapt.R:118-133 and R/kernel_ram.R:129-152 are ONE loop: `i > 2`, the mean of the call's rows
    before the first adaptation, eta(i, k) and `i %% freq` read the CALL's step, whatever the window.  Windows of 32 / 96 steps
    cut calls of a few hundred steps 3 - 12 times: the oracle's bits -- warm-up ending inside a later window (so the first
    running mean comes from the carried row sum), burn-in and thinning that do not divide the window, kernel_ram's freq = 3,
    `until` inside a window, a second call continuing the first; register-row owners, the owners with their matrices in
    LDS (a fixed parameter), the streamed MFMA evaluation (n = 12,001), one chain per workgroup."""
    from fmcmc_amd import _abi as abi
    set_knob(monkeypatch, "window", window)
    if shape == "spec-lat1":
        set_knob(monkeypatch, "lat", "1")
    n = 12001 if shape == "mfma-adaptive" else 3000
    X, y = synth_linreg(n, 3, 20260102)
    init = jitter_init([0, 0, 0, 0, float(np.std(y))], 6, 31)
    init[:, -1] = np.abs(init[:, -1])
    kw = dict(fixed=[False, True, False, False, False]) if shape == "spec-lds-owner" else {}
    if kind == "adapt":
        run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, 5, init, nsteps=333, burnin=41, thin=7, calls=2, warmup=70, until=500.0, **kw)
        run_both(E, O, O.FAM_LINREG, X, y, O.K_ADAPT, 5, init, nsteps=150, warmup=0, **kw)   # (adapting from step 3 of the call)
    else:
        run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, 5, init, nsteps=333, burnin=41, thin=7, calls=2, freq=3, warmup=50, **kw)
        run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, 5, init, nsteps=150, **kw)
    assert abi.last_kernel() == {"spec": "spec", "spec-lds-owner": "spec", "mfma-adaptive": "mfma-adaptive", "spec-lat1": "spec-lat1"}[shape]


def test_full_size_headline_properties(E, monkeypatch):
    """BASELINE configs[1] at its full size (1024 chains x 10,000 iterations, n = 10,000, k = 5; the oracle would need
    minutes), through properties that do not depend on the size: (a) two shards of 512 chains with their chain_base give
    the bits of the single launch; (b) every row is the proposal if its accept bit is set and the previous row otherwise
    (R/mcmc.R:770-778), row 1 is the initial state; (c) accept_count is the population count of the accept bits;
    (d) the headline (MFMA) kernel and the wave-specialised VALU kernel agree bit for bit on all outputs."""
    import torch
    from fmcmc_amd import _abi as abi
    import bench
    chains, iters, k = 1024, 10000, 5
    X, y, init = bench.Config("c2").workload(chains, 0)
    gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
    gk = E.KernelSpec(abi.KERNEL_NORMAL, k, np.zeros(k), np.full(k, 0.02), np.full(k, -E.DBL_MAX),
                      np.full(k, E.DBL_MAX), np.zeros(k, np.uint8))

    def launch(lo, hi):
        st = E.ChainState(init[lo:hi], k)
        r = E.sweep(gm, gk, st, iters, seed=bench.CHAIN_SEED, chain_base=lo, check=True)
        torch.cuda.synchronize()
        return r, st

    full, st_full = launch(0, chains)
    assert abi.last_kernel() == "mfma"
    # (a) sharding
    for lo, hi in ((0, 512), (512, 1024)):
        part, st_part = launch(lo, hi)
        for name in ("samples", "logpost", "draws", "accept_count", "accept_bits"):
            assert torch.equal(getattr(part, name), getattr(full, name)[lo:hi]), name
        assert torch.equal(st_part.theta0, st_full.theta0[lo:hi])
        del part
    # (b) structure of the chain
    bits = full.accept_bits.view(torch.int32)                                  # [C][ceil(nsteps / 32)], bit r <-> row r
    r_idx = torch.arange(iters, device=bits.device)
    acc = ((bits[:, (r_idx >> 5)] >> (r_idx & 31)) & 1).bool()                 # [C][iters]
    assert not bool(acc[:, 0].any())
    s64, d64 = full.samples.view(torch.int64), full.draws.view(torch.int64)   # [C][k][S], compared as bit patterns
    assert torch.equal(s64[:, :, 0], torch.as_tensor(init, device=bits.device).view(torch.int64))
    expect = torch.where(acc[:, None, 1:], d64[:, :, 1:], s64[:, :, :-1])
    assert torch.equal(s64[:, :, 1:], expect)
    # (c) counts
    assert torch.equal(acc.sum(dim=1), full.accept_count.to(acc.sum(dim=1).dtype))
    rate = float(full.accept_count.double().mean().item()) / (iters - 1)
    assert 0.45 < rate < 0.65                                                  # the config's frozen scale: ~0.56
    # (d) a second implementation of the same sweep
    set_knob(monkeypatch, "mfma", "0")
    other, _ = launch(0, chains)
    assert abi.last_kernel() == "spec"
    for name in ("samples", "logpost", "draws", "accept_count", "accept_bits"):
        assert torch.equal(getattr(other, name), getattr(full, name)), name


def test_full_size_logistic_and_wide_properties(E, monkeypatch):
    """BASELINE configs C5 (1024 chains per GPU, logistic n = 100,000, k = 6, kernel_normal_reflective, thin 10) and C4
    (512 chains per GPU, n = 10,000, k = 50, kernel_ram) at their per-GPU sizes, through size-independent properties:
    shards with their chain_base give the bits of the single launch, a thinned run is every thin-th row of the unthinned
    one, every sample respects the reflective bounds."""
    import torch
    from fmcmc_amd import _abi as abi
    big = E.DBL_MAX
    # ---- C5
    rng = np.random.default_rng(20260105)
    n5, C5, k5, steps5 = 100000, 1024, 6, 600
    X5 = rng.standard_normal((n5, 5)); b5 = np.array([-1, .5, -.5, .25, -.25, 1.0])
    y5 = (rng.uniform(size=n5) < 1 / (1 + np.exp(-(b5[0] + X5 @ b5[1:])))).astype(np.float64)
    init5 = b5[None, :] + 0.01 * rng.standard_normal((C5, k5))
    gm = E.DeviceModel(abi.FAM_LOGISTIC, X5, y5, intercept=True, guard=False, prior_div=8.0)
    o = np.ones(k5)
    gk = E.KernelSpec(abi.KERNEL_NORMAL_REFLECTIVE, k5, 0 * o, 0.01 * o, -5 * o, 5 * o, np.zeros(k5, np.uint8))

    def run5(lo, hi, thin, nsteps=steps5, state=None):
        st = state if state is not None else E.ChainState(init5[lo:hi], k5)
        r = E.sweep(gm, gk, st, nsteps, thin=thin, seed=1215, chain_base=lo, check=True)
        torch.cuda.synchronize()
        return r, st

    full, _ = run5(0, C5, 10)
    assert abi.last_kernel() == "logistic-shadow"
    assert full.samples.shape[-1] == steps5 // 10
    assert bool((full.samples.abs() <= 5.0).all())
    part, _ = run5(256, 768, 10)
    assert torch.equal(part.samples, full.samples[256:768]) and torch.equal(part.logpost, full.logpost[256:768])
    dense, _ = run5(0, 256, 1)
    assert torch.equal(dense.samples[:, :, 9::10], full.samples[:256])        # kept rows: 10, 20, ... (R/mcmc.R:786-813)
    assert torch.equal(dense.logpost[:, 9::10], full.logpost[:256])
    # ---- C4
    rng = np.random.default_rng(20260104)
    n4, C4, k4, steps4 = 10000, 512, 50, 240
    X4 = rng.standard_normal((n4, k4 - 2)); b4 = rng.standard_normal(k4 - 1)
    y4 = b4[0] + X4 @ b4[1:] + 2 * rng.standard_normal(n4)
    init4 = np.concatenate([b4, [2.0]])[None, :] + 0.01 * rng.standard_normal((C4, k4)); init4[:, -1] = np.abs(init4[:, -1])
    gm4 = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X4, y4)
    o = np.ones(k4)
    gk4 = E.KernelSpec(abi.KERNEL_RAM, k4, 0 * o, o, -big * o, big * o, np.zeros(k4, np.uint8))
    st1 = E.ChainState(init4, k4)
    one = E.sweep(gm4, gk4, st1, steps4, seed=7, check=True)
    torch.cuda.synchronize()
    # (a continued kernel_ram sweep is NOT the single sweep: eta = k i^(-2/3) uses the loop index of the call, as
    #  R/kernel_ram.R does with env$i -- so C4's size-independent property is sharding: 256 of the chains on their own,
    #  which is also a different instantiation, one chain per workgroup instead of two)
    for lo, hi in ((0, 256), (256, 512)):
        st2 = E.ChainState(init4[lo:hi], k4)
        part = E.sweep(gm4, gk4, st2, steps4, seed=7, chain_base=lo, check=True)
        torch.cuda.synchronize()
        assert torch.equal(part.samples, one.samples[lo:hi]) and torch.equal(part.logpost, one.logpost[lo:hi])
        assert torch.equal(part.accept_count, one.accept_count[lo:hi])
        assert torch.equal(st2.Sigma, st1.Sigma[lo:hi]) and torch.equal(st2.abs_iter, st1.abs_iter[lo:hi])


def test_c4_exact_shape_equals_the_oracle(E, O):
    """BASELINE configs[3] at EXACTLY its per-GPU shape -- 512 chains, n = 10,000, 48 covariates (k = 50), kernel_ram -- against
    the oracle for 24 steps (the oracle needs a few seconds for 512 x 24 evaluations of 10,000 x 48): every output, the accept
    bitmap, the adapted factors.  The dataflow kernel with its compile-time K-block count (12) is what runs."""
    import bench
    from fmcmc_amd import _abi as abi
    cfg = bench.Config("c4")
    X, y, init = cfg.workload(cfg.chains, 0)
    assert X.shape == (10000, 48) and init.shape == (512, 50)
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, 50, init, nsteps=24, seed=bench.CHAIN_SEED)
    assert abi.last_kernel() == "wide-dataflow"


@pytest.mark.parametrize("case", range(int(os.environ.get("FMCMC_TEST_RANDOM_CASES2", "80"))))   # soak: 2000 passed (round 5, last code: 6000 -- case 1566 found a launch without a loop, see test_logistic_fixed_parameter_beyond...)
def test_randomised_option_cases(E, O, case):
    """Second randomised sweep, over the options the first one leaves at their defaults: unguarded log-posteriors,
    flat / Gaussian priors, non-zero proposal means, explicit update sequences, kernel_ram's constr mask and target rate,
    kernel_adapt's window / stride / until / eps / Sd, wide models up to k = 64, two or three consecutive calls."""
    rng = np.random.default_rng(5150000 + case)
    fam = str(rng.choice(["linreg", "linreg", "logistic"]))
    n = int(rng.choice([3, 64, 500, 777, 1024, 1500, 4097, 10000]))
    p = int(rng.choice([1, 2, 3, 5, 6, 9, 12, 20, 33, 47, 62])) if fam == "linreg" else int(rng.choice([1, 2, 4, 6, 8, 11]))
    intercept = bool(rng.integers(0, 2))
    chains = int(rng.choice([1, 2, 4, 6, 17, 128, 256]))
    k = p + int(intercept) + (1 if fam == "linreg" else 0)
    if k > 64:
        p -= k - 64; k = 64
    kind_name = str(rng.choice(["normal", "reflective", "adapt", "ram", "unif"]))
    nsteps = int(max(8, min(70, 1.2e8 / (chains * n * p * 2))))
    calls = int(rng.integers(1, 4))
    beta = rng.uniform(-1.0, 1.0, p + 1)
    X = rng.standard_normal((n, p))
    if fam == "linreg":
        y = (beta[0] if intercept else 0.0) + X @ beta[1:] + 1.5 * rng.standard_normal(n)
        base = list(beta[(0 if intercept else 1):]) + [1.5]
        famc, kw = O.FAM_LINREG, dict(intercept=intercept, guard=bool(rng.integers(0, 2)))
        lb = [-30.0] * (k - 1) + [0.2]
    else:
        eta = (beta[0] if intercept else 0.0) + X @ beta[1:]
        y = (rng.uniform(size=n) < 1 / (1 + np.exp(-eta))).astype(np.float64)
        base = list(beta[(0 if intercept else 1):])
        famc, kw = O.FAM_LOGISTIC, dict(intercept=intercept, prior_div=float(rng.choice([0.0, 8.0, 2.5])))
        lb = [-5.0] * k
    init = jitter_init(base, chains, 11 + case)
    if fam == "linreg":
        init[:, -1] = np.abs(init[:, -1]) + 0.3
    fixed = np.zeros(k, bool)
    if k > 3 and rng.random() < 0.4:
        fixed[rng.choice(k - 1, size=int(rng.integers(1, 3)), replace=False)] = True
    free = np.nonzero(~fixed)[0]
    opts = dict(fixed=fixed)
    if kind_name in ("normal", "reflective"):
        kind = O.K_NORMAL if kind_name == "normal" else O.K_NORMAL_REFLECTIVE
        opts.update(scale=rng.uniform(0.002, 0.02, k), mu=rng.uniform(-0.002, 0.002, k))
        r = rng.random()
        if r < 0.25:
            opts["scheme"] = list(rng.permutation(free) + 1)          # explicit sequence, 1-based as in R
        elif r < 0.4:
            opts["scheme"] = "random"
        if kind_name == "reflective":
            opts.update(lb=lb, ub=30.0 if fam == "linreg" else 5.0)
    elif kind_name == "unif":
        kind = O.K_UNIF_REFLECTIVE if case % 2 else O.K_UNIF
        opts.update(min_=-0.01, max_=0.012)
        if case % 2:
            opts.update(lb=lb, ub=30.0 if fam == "linreg" else 5.0)
    elif kind_name == "adapt":
        kind = O.K_ADAPT
        warm = int(rng.integers(2, 12))
        opts.update(warmup=warm, eps=float(rng.choice([1e-4, 1e-6])), freq=int(rng.choice([1, 1, 2, 3])))
        if rng.random() < 0.3:
            opts.update(bw=int(rng.integers(3, 8)), Sd=0.7)
            opts["warmup"] = max(warm, opts["bw"])                  # kernel_adapt stops when bw > warmup
        if rng.random() < 0.3:
            opts["until"] = float(nsteps // 2)
        if rng.random() < 0.4:
            opts.update(lb=lb, ub=30.0 if fam == "linreg" else 5.0)
    else:
        kind = O.K_RAM
        opts.update(arate=float(rng.choice([0.234, 0.4])), warmup=int(rng.choice([0, 0, 4])), freq=int(rng.choice([1, 1, 2])))
        if rng.random() < 0.3:
            M = (np.abs(np.subtract.outer(np.arange(k), np.arange(k))) <= 2).astype(float)
            opts["constr"] = M
        if rng.random() < 0.3:
            opts.update(lb=lb, ub=30.0 if fam == "linreg" else 5.0)
    run_both(E, O, famc, X, y, kind, k, init, nsteps=nsteps, burnin=int(rng.integers(0, 3)), thin=int(rng.choice([1, 1, 2])),
             seed=int(rng.integers(1, 10**6)), chain_base=int(rng.choice([0, 3, 4096])), calls=calls, **kw, **opts)


@pytest.mark.parametrize("case", range(int(os.environ.get("FMCMC_TEST_RANDOM_CASES3", "64"))))   # soak: 4000 + 1500 (with region E) passed on the final code of round 4
def test_randomised_round4_regions(E, O, monkeypatch, case):
    """Third randomised sweep, over the regions round 4 opened and the first two never draw: (A) linreg beyond the operand
    registers of the MFMA kernel (10,240 < n <= 45,000 at p <= 3, 5,120 < n at p <= 7: mfma-streamed / mfma-adaptive), (B) 64 < k
    <= 128 (big-k), (C) logistic regression on the observation-sharded evaluation at ragged n, p <= 8, chain counts around 512,
    (D) the wave-specialised kernel at any n <= 10,240 with p <= 7, (E) few chains on long data (the long-data form; linreg narrow
    and wide, logistic) -- each with random kernels, bounds, fixed masks, burn-in,
    thinning and one or two calls.  Whatever kernel the dispatcher picks: the oracle's bits."""
    from fmcmc_amd import _abi as abi
    rng = np.random.default_rng(8640000 + case)
    region = "ABCDE"[case % 5]
    intercept = bool(rng.integers(0, 4))            # mostly with an intercept
    calls = int(rng.integers(1, 3))
    fam = "logistic" if region == "C" else "linreg"     # (region E draws its own)
    if region == "A":
        p = int(rng.choice([1, 2, 3, 3, 4, 5, 7, 8, 10, 13, 14]))
        n = int(rng.integers(10241, 45001)) if p <= 3 else (int(rng.integers(5121 if p <= 5 else 4097, 20001)) if p <= 7 else int(rng.integers(513, 12001)))
        chains = int(rng.choice([1, 4, 5, 9]))
        kind_name = str(rng.choice(["normal", "reflective", "unif", "adapt", "ram"]))
    elif region == "B":
        k_ = int(rng.integers(65, 129))
        p = k_ - 1 - int(intercept)
        n = int(rng.integers(200, 2500))
        chains = int(rng.choice([1, 2, 3, 5]))
        kind_name = str(rng.choice(["normal", "reflective", "unif", "adapt", "ram"]))
    elif region == "C":
        p = int(rng.integers(1, 17))
        n = int(rng.integers(1024, 24001))
        chains = int(rng.choice([3, 8, 130, 511, 513, 600]))
        kind_name = str(rng.choice(["normal", "reflective"]))
        set_knob(monkeypatch, "shard", str(rng.choice(["1", "1", "0"])))
    elif region == "E":     # few chains on long data: the long-data form, linreg narrow / wide and logistic
        fam = str(rng.choice(["linreg", "linreg", "logistic"]))
        chains = int(rng.choice([1, 2, 4, 7]))
        if fam == "logistic":
            p = int(rng.integers(1, 17))
            n = int(rng.integers(20000, 120001))
        elif rng.random() < 0.3:
            p = int(rng.integers(16, 41))
            n = int(rng.integers(24577, 60001))
        else:
            p = int(rng.choice([1, 2, 3, 5, 8, 14]))
            n = int(rng.integers(30000, 150001))
        kind_name = str(rng.choice(["normal", "reflective", "adapt", "ram"]))
    else:
        p = int(rng.integers(1, 8))
        n = int(rng.integers(513, 10241 if p <= 3 else (5121 if p <= 5 else 4097)))
        chains = int(rng.choice([1, 3, 4, 6, 13]))
        kind_name = str(rng.choice(["adapt", "ram", "normal"]))
    k = p + int(intercept) + (1 if fam == "linreg" else 0)
    nsteps = int(max(7, min(60, 1.5e8 / (chains * n * max(p, 1) * 2 * calls))))
    beta = rng.uniform(-1.0, 1.0, p + 1) * (0.3 if region == "B" else 1.0)
    X = rng.standard_normal((n, p)) * (0.4 if region == "B" else 1.0)
    if fam == "linreg":
        y = (beta[0] if intercept else 0.0) + X @ beta[1:] + 1.5 * rng.standard_normal(n)
        base = list(beta[(0 if intercept else 1):]) + [1.5]
        famc, kw = O.FAM_LINREG, dict(intercept=intercept, guard=bool(rng.integers(0, 4)))
        lb, ub = [-30.0] * (k - 1) + [0.2], 30.0
    else:
        eta = (beta[0] if intercept else 0.0) + X @ beta[1:]
        y = (rng.uniform(size=n) < 1 / (1 + np.exp(-eta))).astype(np.float64)
        base = list(beta[(0 if intercept else 1):])
        famc, kw = O.FAM_LOGISTIC, dict(intercept=intercept, prior_div=float(rng.choice([0.0, 8.0])))
        lb, ub = [-5.0] * k, 5.0
    init = jitter_init(base, chains, 21 + case)
    if fam == "linreg":
        init[:, -1] = np.abs(init[:, -1]) + 0.3
    fixed = np.zeros(k, bool)
    if k > 3 and rng.random() < 0.35:
        fixed[rng.choice(k - 1, size=int(rng.integers(1, 3)), replace=False)] = True
    opts = dict(fixed=fixed)
    if kind_name in ("normal", "reflective"):
        kind = O.K_NORMAL if kind_name == "normal" else O.K_NORMAL_REFLECTIVE
        opts.update(scale=rng.uniform(0.002, 0.02, k))
        if kind_name == "reflective":
            opts.update(lb=lb, ub=ub)
    elif kind_name == "unif":
        kind = O.K_UNIF_REFLECTIVE if case % 8 >= 4 else O.K_UNIF
        opts.update(min_=-0.01, max_=0.012)
        if kind == O.K_UNIF_REFLECTIVE:
            opts.update(lb=lb, ub=ub)
    elif kind_name == "adapt":
        kind = O.K_ADAPT
        opts.update(warmup=int(rng.integers(2, 10)), eps=float(rng.choice([1e-4, 1e-6])))
        if region != "B" and rng.random() < 0.3:
            opts["freq"] = 2
        if rng.random() < 0.3:
            opts["until"] = float(nsteps // 2)
        if rng.random() < 0.4:
            opts.update(lb=lb, ub=ub)
    else:
        kind = O.K_RAM
        opts.update(arate=float(rng.choice([0.234, 0.4])), warmup=int(rng.choice([0, 0, 3])))
        if rng.random() < 0.3:
            opts["constr"] = (np.abs(np.subtract.outer(np.arange(k), np.arange(k))) <= 2).astype(float)
        if rng.random() < 0.35:
            opts.update(lb=lb, ub=ub)
    burnin = int(rng.integers(0, 3))
    thin = int(rng.choice([1, 1, 2, 3]))
    if burnin + thin > nsteps - 1:
        burnin, thin = 0, 1
    run_both(E, O, famc, X, y, kind, k, init, nsteps=nsteps, burnin=burnin, thin=thin, seed=int(rng.integers(1, 10**6)),
             chain_base=int(rng.choice([0, 7, 4096])), calls=calls, **kw, **opts)
    _R4_PICKED.setdefault(region, set()).add(abi.last_kernel())


_R4_PICKED = {}


def test_randomised_round4_regions_reached_their_kernels():
    """(runs after the cases above) the regions were drawn to reach the round-4 kernels -- say which ones they did reach"""
    import torch
    if not _R4_PICKED or torch.cuda.get_device_properties(0).multi_processor_count < 256:
        pytest.skip("cases not run in this process / not a 256-CU device")
    if int(os.environ.get("FMCMC_TEST_RANDOM_CASES3", "64")) >= 64:
        assert {"mfma-streamed", "mfma-adaptive"} <= _R4_PICKED["A"], _R4_PICKED
        assert _R4_PICKED["B"] == {"big-k"}, _R4_PICKED
        assert {"logistic-sharded", "logistic-shadow"} & _R4_PICKED["C"], _R4_PICKED
        assert any(kn.startswith("spec") for kn in _R4_PICKED["D"]), _R4_PICKED
        assert "long-sharded" in _R4_PICKED["E"], _R4_PICKED


@pytest.mark.parametrize("kind_name", ["normal", "ram"])
def test_sharded_evaluation_in_consecutive_launches(E, O, monkeypatch, kind_name):
    """More than 512 chains per GPU on a wide model: the sweep runs as consecutive cooperative launches of 512 chains
    (chain windows of one call: every per-chain array advanced, RNG ids continued) -- the oracle's bits for all 1024
    chains, continued over two calls, and the bits of the chain-sharded kernel."""
    import torch
    from fmcmc_amd import _abi as abi
    set_knob(monkeypatch, "shard", "1")
    n, p, chains = 1500, 18, 1024
    rng = np.random.default_rng(31)
    beta = rng.uniform(-1.0, 1.0, p + 1)
    X = rng.standard_normal((n, p))
    y = beta[0] + X @ beta[1:] + 2.0 * rng.standard_normal(n)
    init = jitter_init(list(beta) + [2.0], chains, 6)
    init[:, -1] = np.abs(init[:, -1]) + 0.1
    kind = O.K_NORMAL if kind_name == "normal" else O.K_RAM
    opts = dict(scale=0.01, fixed=[False] * 3 + [True] + [False] * (p - 2)) if kind_name == "normal" else {}
    a, _ = run_both(E, O, O.FAM_LINREG, X, y, kind, p + 2, init, nsteps=16, burnin=1, thin=2, calls=2, chain_base=40, **opts)
    if torch.cuda.get_device_properties(0).multi_processor_count >= 256:
        assert abi.last_kernel() in SHARDED
    set_knob(monkeypatch, "shard", "0")
    b, _ = run_both(E, O, O.FAM_LINREG, X, y, kind, p + 2, init, nsteps=16, burnin=1, thin=2, calls=2, chain_base=40, **opts)
    assert abi.last_kernel() in ("streamed-wide", "streamed")
    assert _bits_equal(a.samples.cpu().numpy(), b.samples.cpu().numpy())
