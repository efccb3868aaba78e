"""GPU tests of the drop-in API layer and of the strongest parity statements:
 * the engine, fed R's own Mersenne-Twister draw stream, reproduces the outputs fmcmc PRINTS (README);
 * MCMC(..., conv_checker = convergence_gelman()) stops at the same bulk with the same R-hat history as
   the oracle's restatement of MCMC_with_conv_checker;
 * the committed golden vectors of the canonical stream."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import synth_linreg

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "readme_goldens.json")))


@pytest.fixture(scope="module")
def E():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from fmcmc_amd import engine
    return engine


def sig(x, d):
    return float("%.*g" % (d, x))


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def fed_stream(O, g, nsteps, kz, ram_df=None):
    """R's draw order for one chain (R/mcmc.R:726 then the kernel's draws per step)."""
    logu = np.log(g.runif(nsteps))
    z = np.zeros((nsteps, kz))
    for i in range(1, nsteps):  # rows 1.. hold the variates of loop steps 2..
        z[i] = g.rt(kz, ram_df) if ram_df else g.rnorm(kz)
    return logu, z


def test_fed_replay_reproduces_the_readme_on_the_gpu(E, O, readme_data):
    import torch
    from fmcmc_amd import _abi as abi
    X, y = readme_data
    gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y, intercept=True, guard=True)
    big = E.DBL_MAX
    g = O.RRng(1215)

    def run(kind, scale, init, ram=False):
        logu, z = fed_stream(O, g, 5000, 3, ram_df=3.0 if ram else None)
        gk = E.KernelSpec(kind, 3, np.zeros(3), np.full(3, scale), np.full(3, -big), np.full(3, big),
                          np.zeros(3, np.uint8), warmup=0)
        st = E.ChainState(np.asarray(init, dtype=np.float64)[None, :], 3)
        r = E.sweep(gm, gk, st, 5000, fed_logu=torch.as_tensor(logu[None, :]).cuda().contiguous(),
                    fed_z=torch.as_tensor(z[None, :, :]).cuda().contiguous())
        torch.cuda.synchronize()
        return r, st

    # README.md:156-201 (G1)
    r1, _ = run(abi.KERNEL_NORMAL, 1.0, [0, 0, O.r_sd(y)])
    s = r1.samples.cpu().numpy()[0].T
    assert [sig(v, 4) for v in s.mean(0)] == G["G1"]["mean"]
    assert [sig(v, 4) for v in s.std(0, ddof=1)] == [sig(v, 4) for v in G["G1"]["sd"]]
    q = np.quantile(s[:, 0], [.025, .25, .5, .75, .975])
    assert [sig(v, 4) for v in q] == G["G1"]["q_par1"]
    assert list(O.accept_steps(r1.accept_bits.cpu().numpy().view(np.uint32)[0])) == [
        3, 5, 8, 10, 14, 32, 67, 544, 786, 834, 1598, 2764, 3693, 3826, 4039, 4514, 4613, 4776, 4898, 4916, 4950]
    # README.md:209-246 (G4): kernel_normal(scale=.05) from the last row, then kernel_ram()
    r2, _ = run(abi.KERNEL_NORMAL, 0.05, s[-1])
    s2 = r2.samples.cpu().numpy()[0].T
    assert int(r2.accept_count[0]) == 3641
    r3, st3 = run(abi.KERNEL_RAM, 1.0, s2[-1], ram=True)
    assert int(r3.accept_count[0]) == 1761
    assert sig(1761 / 4999, 7) == G["G4"]["ram_accept_rate"]
    assert np.allclose(st3.Sigma.cpu().numpy()[0], [[0.18001104, 0, 0], [0.01481571, 0.17576006, 0],
                                                    [0.00806067, -0.00281060, 0.11951173]], atol=5e-9)


def test_committed_golden_vectors(E, O):
    from golden.make_philox_vectors import cases, make_inputs, kernel_kwargs, hexbits
    from fmcmc_amd import _abi as abi
    import torch
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "philox_vectors.json")))
    for name, spec in cases().items():
        X, y, init = make_inputs(spec)
        kind, kw = kernel_kwargs(O, spec)
        ok = O.Kernel(kind, spec["p"] + 2, **kw)
        gk = E.KernelSpec(kind, ok.k, ok.mu, ok.scale, ok.lb, ok.ub, ok.fixed, scheme=ok.scheme, freq=ok.freq, warmup=ok.warmup,
                          eps=ok.eps, arate=ok.arate, scheme_seq=ok.scheme_seq, constr=ok.constr)
        st = E.ChainState(init, ok.kf)
        r = E.sweep(E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y), gk, st, spec["nsteps"], seed=spec["seed"])
        torch.cuda.synchronize()
        samples = r.samples.cpu().numpy().transpose(0, 2, 1)
        draws = r.draws.cpu().numpy().transpose(0, 2, 1)
        got = {"accept_count": [int(v) for v in r.accept_count.cpu().numpy()],
               "accept_bits": [int(v) for v in r.accept_bits.cpu().numpy().view(np.uint32).ravel()],
               "last_row": hexbits(samples[:, -1, :]), "logpost_last": hexbits(r.logpost.cpu().numpy()[:, -1]),
               "sample_checksum": hexbits([samples.sum(), np.abs(draws).sum()])}
        for key in ("accept_count", "accept_bits", "last_row", "logpost_last"):
            assert got[key] == fx[name][key], (name, key)


def test_gelman_partial_kernel(E, O):
    import torch
    from fmcmc_amd import _abi as abi
    from test_abi import numpy_gelman_partial
    rng = np.random.default_rng(3)
    Cn, k, S = 37, 6, 500
    x = rng.standard_normal((Cn, k, S)) * 0.5 + rng.standard_normal((Cn, k, 1)) * 0.2 + 3.0
    cols = np.array([0, 2, 3, 5], dtype=np.int32)
    row0, N = 250, 250
    xd = torch.as_tensor(x).cuda()
    cd = torch.as_tensor(cols).cuda()
    center = torch.as_tensor(x[0, cols, row0]).cuda()
    L = abi.lib()
    p = len(cols)
    part = torch.zeros(int(L.fmcmc_gelman_partial_len(p)), dtype=torch.float64, device="cuda")
    work = torch.empty(int(L.fmcmc_gelman_work_len(Cn, p)), dtype=torch.float64, device="cuda")
    rc = L.fmcmc_gelman_partial_dev(xd.data_ptr(), Cn, k, S, row0, N, cd.data_ptr(), p, center.data_ptr(),
                                    work.data_ptr(), part.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    win = x[:, cols, row0:].transpose(0, 2, 1)  # [m][N][p]
    ref = numpy_gelman_partial(win, x[0, cols, row0])
    assert np.allclose(part.cpu().numpy(), ref, rtol=1e-10, atol=1e-12)
    psrf = np.empty(p); mps = C.c_double(); dp = C.POINTER(C.c_double)
    pn = part.cpu().numpy()
    assert L.fmcmc_gelman_finish(pn.ctypes.data_as(dp), p, N, psrf.ctypes.data_as(dp), C.byref(mps)) == 0
    opsrf, ompsrf = O.gelman(win)
    assert np.allclose(psrf, opsrf, rtol=1e-9) and abs(mps.value - ompsrf) < 1e-9 * ompsrf


@pytest.mark.parametrize("Cn,k,S,row0,ncols", [(512, 50, 6000, 1000, 50), (9, 20, 333, 101, 17), (5, 40, 90, 7, 33)])
def test_gelman_partial_kernel_wide(E, O, Cn, k, S, row0, ncols):
    """The device reduction of convergence_gelman at config C4's width (512 chains x p = 50 x a 5,000-row window: the
    v_mfma_f64_16x16x4 rank-N update) and at ragged shapes (1 to 3 column blocks, windows that are no multiple of the
    K-step, a column subset): partial vector == its numpy definition, psrf / mpsrf == the oracle's coda restatement
    (R/convergence.R:191-246 -> coda::gelman.diag)."""
    import torch
    from fmcmc_amd import _abi as abi
    from test_abi import numpy_gelman_partial
    g = torch.Generator(device="cuda"); g.manual_seed(Cn + k)
    xd = torch.randn((Cn, k, S), dtype=torch.float64, device="cuda", generator=g) * 0.5 + 3.0
    xd += torch.randn((Cn, k, 1), dtype=torch.float64, device="cuda", generator=g) * 0.05        # chains differ a little
    xd += torch.cumsum(torch.randn((Cn, k, S), dtype=torch.float64, device="cuda", generator=g), 2) * 0.01   # autocorrelated
    N = S - row0
    cols = np.sort(np.random.default_rng(k).choice(k, size=ncols, replace=False)).astype(np.int32)
    cd = torch.as_tensor(cols).cuda()
    center = xd[0, cd.long(), row0].contiguous()
    L = abi.lib()
    p = ncols
    part = torch.zeros(int(L.fmcmc_gelman_partial_len(p)), dtype=torch.float64, device="cuda")
    work = torch.empty(int(L.fmcmc_gelman_work_len(Cn, p)), dtype=torch.float64, device="cuda")
    rc = L.fmcmc_gelman_partial_dev(xd.data_ptr(), Cn, k, S, row0, N, cd.data_ptr(), p, center.data_ptr(),
                                    work.data_ptr(), part.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    win = xd[:, cd.long(), row0:].cpu().numpy().transpose(0, 2, 1)  # [m][N][p]
    ref = numpy_gelman_partial(win, center.cpu().numpy())
    got = part.cpu().numpy()
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-11 * np.abs(ref).max())
    psrf = np.empty(p); mps = C.c_double(); dp = C.POINTER(C.c_double)
    assert L.fmcmc_gelman_finish(got.ctypes.data_as(dp), p, N, psrf.ctypes.data_as(dp), C.byref(mps)) == 0
    opsrf, ompsrf = O.gelman(win)
    assert np.allclose(psrf, opsrf, rtol=1e-9) and abs(mps.value - ompsrf) < 1e-9 * ompsrf


def test_mcmc_api_end_to_end(E, O, readme_data):
    import fmcmc_amd as f
    X, y = readme_data
    fun = f.gaussian_linreg(X, y)
    kern = f.kernel_normal(scale=0.05)
    ans = f.MCMC([0, 0, O.r_sd(y)], fun, 2000, seed=1215, kernel=kern)
    assert isinstance(ans, f.Mcmc) and ans.mcpar == (1, 2000, 1) and ans.data.shape == (2000, 3)
    ro = O.run(O.Model(O.FAM_LINREG, X, y), O.Kernel(O.K_NORMAL, 3, scale=0.05), [0, 0, O.r_sd(y)], nsteps=2000, seed=1215)
    assert np.array_equal(_bits(ans.data), _bits(ro.samples[0]))
    assert np.array_equal(_bits(f.get_logpost()), _bits(ro.logpost[0]))
    assert np.array_equal(_bits(f.get_draws()), _bits(ro.draws[0]))
    # restart from the previous result with burn-in and thinning (R/mcmc.R:344-378)
    ans2 = f.MCMC(ans, fun, 1000, seed=7, kernel=f.kernel_adapt(warmup=100), burnin=100, thin=10)
    assert ans2.mcpar == (110, 1000, 10) and ans2.niter == 90
    assert np.isfinite(ans2.data).all() and abs(ans2.data[:, 0].mean() - 3.1) < 0.5
    # several chains, kernel state persists in the kernel object like in R (vignette workflow)
    kr = f.kernel_ram()
    l1 = f.MCMC(np.tile([3, 2, 4.0], (4, 1)), fun, 500, seed=3, nchains=4, kernel=kr)
    assert isinstance(l1, f.McmcList) and l1.nchain == 4 and list(kr.abs_iter) == [499] * 4
    l2 = f.MCMC(l1, fun, 500, seed=3, nchains=4, kernel=kr)
    assert list(kr.abs_iter) == [998] * 4 and kr[2].Sigma.shape == (3, 3) and len(f.get_logpost()) == 4
    with pytest.raises(RuntimeError, match="undefined"):
        f.MCMC([0, 0, 0.05], f.gaussian_linreg(X, y, guard=False), 300, seed=1, kernel=f.kernel_normal(scale=1.0))


def test_mcmc_autostop_matches_the_oracle(E, O, readme_data):
    """MCMC_with_conv_checker + convergence_gelman (R/mcmc.R:841-1019, R/convergence.R:191-246)."""
    import fmcmc_amd as f
    X, y = readme_data
    init = np.tile([0, 0, O.r_sd(y)], (4, 1)) + 0.3 * np.random.default_rng(2).standard_normal((4, 3))
    chk = f.convergence_gelman(200, threshold=1.10)
    ans = f.MCMC(init, f.gaussian_linreg(X, y), 5000, seed=99, nchains=4, kernel=f.kernel_normal(scale=0.05),
                 conv_checker=chk)
    ro = O.mcmc_with_conv_checker(O.Model(O.FAM_LINREG, X, y), O.Kernel(O.K_NORMAL, 3, scale=0.05), init, 5000, 4, 200,
                                  seed=99)
    assert [h[0] for h in chk.history] == [h[0] for h in ro.history]
    assert np.allclose([h[1] for h in chk.history], [h[1] for h in ro.history], rtol=1e-8)
    assert ans.niter == ro.samples.shape[1] and ans.nchain == 4
    assert np.array_equal(_bits(ans.as_array()), _bits(ro.samples))
    assert list(ans.iters) == list(ro.iters)


def test_mcmc_with_more_than_64_parameters_and_the_gelman_checker(E, O):
    """MCMC(..., kernel_ram(), conv_checker = convergence_gelman()) at k = 70 (round 4: FMCMC_MAX_K 128): the chains come from
    mh_sweep_bigk, the checker's window statistics from the host-side torch path above 64 columns -- the R-hat history, the
    stop bulk and the samples equal the oracle's (R/mcmc.R:841-1019, R/convergence.R:191-246)."""
    import fmcmc_amd as f
    from fmcmc_amd import _abi as abi
    k = 70
    X, y = synth_linreg(600, k - 2, 11, beta=np.linspace(1.0, -1.0, k - 1), sigma=2.0)
    init = np.tile(np.r_[np.linspace(1.0, -1.0, k - 1), 2.0], (3, 1)) + 0.02 * np.random.default_rng(3).standard_normal((3, k))
    init[:, -1] = np.abs(init[:, -1])
    chk = f.convergence_gelman(100, threshold=1.5)
    ans = f.MCMC(init, f.gaussian_linreg(X, y), 600, seed=7, nchains=3, kernel=f.kernel_ram(), conv_checker=chk)
    assert abi.last_kernel() == "big-k"
    ro = O.mcmc_with_conv_checker(O.Model(O.FAM_LINREG, X, y), O.Kernel(O.K_RAM, k), init, 600, 3, 100, seed=7, threshold=1.5)
    assert [h[0] for h in chk.history] == [h[0] for h in ro.history] and len(chk.history) >= 1
    assert np.allclose([h[1] for h in chk.history], [h[1] for h in ro.history], rtol=1e-7)
    assert np.array_equal(_bits(ans.as_array()), _bits(ro.samples))


def test_logistic_and_large_k_ram(E, O):
    """BASELINE configs 4 and 5 in miniature: kernel_ram at k = 50, logistic with reflective bounds."""
    from test_gpu_parity import run_both, jitter_init
    rng = np.random.default_rng(8)
    n, p = 400, 48
    X = rng.standard_normal((n, p)); beta = rng.standard_normal(p + 1)
    y = beta[0] + X @ beta[1:] + 2.0 * rng.standard_normal(n)
    init = jitter_init(np.concatenate([beta, [2.0]]), 3, 4)
    init[:, -1] = np.abs(init[:, -1])
    run_both(E, O, O.FAM_LINREG, X, y, O.K_RAM, p + 2, init, nsteps=120, calls=2)
    n, p = 3000, 5
    X = rng.standard_normal((n, p)); b = np.array([-1, .5, -.5, .25, -.25, 1.0])
    yb = (rng.uniform(size=n) < 1 / (1 + np.exp(-(b[0] + X @ b[1:])))).astype(np.float64)
    run_both(E, O, O.FAM_LOGISTIC, X, yb, O.K_NORMAL_REFLECTIVE, 6, jitter_init(b, 9, 5), nsteps=150, thin=10,
             prior_div=8.0, scale=0.01, lb=-5.0, ub=5.0)


def test_host_pointer_entry_point(E, O):
    """fmcmc_mcmc_run_host: the call an R `.Call` shim makes (all pointers are host memory)."""
    from fmcmc_amd import _abi as abi
    X, y = synth_linreg(1500, 2, 5)
    Cn, k, nsteps, burnin, thin = 5, 4, 300, 20, 3
    init = np.ascontiguousarray(np.array([0, 0, 0, 4.0])[None, :] + 0.1 * np.random.default_rng(0).standard_normal((Cn, k)))
    for kind, okw in ((O.K_NORMAL, dict(scale=0.05)), (O.K_ADAPT, dict(warmup=50)), (O.K_RAM, {})):
        ok = O.Kernel(kind, k, **okw)
        ro = O.run(O.Model(O.FAM_LINREG, X, y), ok, init, nsteps=nsteps, burnin=burnin, thin=thin, seed=77)
        S = (nsteps - burnin) // thin
        Xc = np.ascontiguousarray(X.T); yc = np.ascontiguousarray(y)
        th = init.copy(); f0 = np.zeros(Cn); abs_iter = np.zeros(Cn, np.int64)
        Sig = np.zeros((Cn, k, k)); mp = np.zeros((Cn, k)); hm = np.zeros(Cn, np.int32); ne = np.zeros(Cn, np.int32)
        samples = np.empty((Cn, k, S)); lp = np.empty((Cn, S)); dr = np.empty((Cn, k, S))
        acc = np.zeros(Cn, np.int64); bits = np.zeros((Cn, (nsteps + 31) // 32), np.uint32)
        status = np.zeros(Cn, np.int32); sstep = np.zeros(Cn, np.int64); stheta = np.zeros((Cn, k))
        P = lambda a: a.ctypes.data
        m = abi.Model(abi.FAM_GAUSSIAN_LINREG, 2, 1500, P(Xc), P(yc), 1, 1, 0.0)
        kk = abi.Kernel(kind, k, P(ok.mu), P(ok.scale), P(ok.lb), P(ok.ub), P(ok.fixed), 0, 1, ok.warmup, 0, ok.until,
                        ok.eps, ok.arate, ok.Sd)
        r = abi.Run(Cn, nsteps, burnin, thin, 77, 0, 0, 0, 0, None, None)
        st = abi.State(P(th), P(f0), P(abs_iter), P(Sig), P(mp), P(hm), P(ne), 1, 0)
        out = abi.Out(P(samples), P(lp), P(dr), P(acc), P(bits), P(status), P(sstep), P(stheta))
        rc = abi.lib().fmcmc_mcmc_run_host(C.byref(m), C.byref(kk), C.byref(r), C.byref(st), C.byref(out), 0)
        assert rc == 0, abi.last_error()
        assert np.array_equal(_bits(samples), _bits(ro.samples_cks)) and np.array_equal(_bits(lp), _bits(ro.logpost))
        assert np.array_equal(_bits(dr), _bits(ro.draws_cks)) and np.array_equal(bits, ro.accept_bits)
        assert np.array_equal(_bits(th), _bits(ro.state.theta0)) and np.array_equal(acc, ro.accept_count)
        if kind != O.K_NORMAL:
            assert np.array_equal(_bits(Sig), _bits(ro.state.Sigma)) and np.array_equal(abs_iter, ro.state.abs_iter)
    # argument errors come back through the return code + fmcmc_last_error()
    r_bad = abi.Run(Cn, 10, 10, 1, 77, 0, 0, 0, 0, None, None)
    assert abi.lib().fmcmc_mcmc_run_host(C.byref(m), C.byref(kk), C.byref(r_bad), C.byref(st), C.byref(out), 0) == abi.ERR_ARG
    assert "-burnin- (10) cannot be >= than -nsteps- (10)." in abi.last_error()


@pytest.mark.parametrize("kind", ["normal", "ram", "ram_qfun_normal", "ram_qfun_t2.5", "unif"])
def test_rng_stream_entry_point_equals_in_library_stream(E, O, kind):
    """fmcmc_rng_stream_dev + rng_mode FED is bit-identical to rng_mode PHILOX (what bench.py relies on), for every variate
    family a kernel draws: N(0,1), kernel_ram's rt(k, k) / rnorm(k) / rt(k, df) (R/kernel_ram.R:68), U(0,1)."""
    import torch
    from fmcmc_amd import _abi as abi
    X, y = synth_linreg(10000, 3, 20260102)
    init = np.array([0, 0, 0, 0, float(np.std(y))])[None, :] + 0.1 * np.random.default_rng(3).standard_normal((6, 5))
    init[:, -1] = np.abs(init[:, -1])
    okw = {"normal": dict(scale=0.02), "ram": {}, "ram_qfun_normal": dict(ram_qfun=1), "ram_qfun_t2.5": dict(ram_qfun=2, ram_df=2.5),
           "unif": dict(min_=-0.03, max_=0.04)}[kind]
    ok = O.Kernel(O.K_NORMAL if kind == "normal" else (O.K_UNIF if kind == "unif" else O.K_RAM), 5, **okw)
    gk = E.KernelSpec(ok.kind, 5, ok.mu, ok.scale, ok.lb, ok.ub, ok.fixed, warmup=ok.warmup, eps=ok.eps, arate=ok.arate,
                      ram_qfun=ok.ram_qfun, ram_df=ok.ram_df, ram_eta_exp=ok.ram_eta_exp)
    gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
    st1, st2 = E.ChainState(init, 5), E.ChainState(init, 5)
    a = E.sweep(gm, gk, st1, 120, seed=11, chain_base=40)
    logu, z = E.rng_stream(st2, gk, 120, seed=11, chain_base=40)
    b = E.sweep(gm, gk, st2, 120, seed=11, chain_base=40, fed_logu=logu, fed_z=z)
    torch.cuda.synchronize()
    assert np.array_equal(_bits(a.samples.cpu().numpy()), _bits(b.samples.cpu().numpy()))
    assert np.array_equal(_bits(a.logpost.cpu().numpy()), _bits(b.logpost.cpu().numpy()))
    ro = O.run(O.Model(O.FAM_LINREG, X, y), ok, init, nsteps=120, seed=11, chain_base=40)
    assert np.array_equal(_bits(b.samples.cpu().numpy()), _bits(ro.samples_cks))


def test_mcmc_api_uniform_kernel_random_scheme_autostop(E, O, readme_data):
    """kernel_unif_reflective + scheme = "random" through MCMC() with convergence_gelman: the update plan belongs to the
    kernel object, so every bulk reuses it (R/kernel.R:106-113) -- chains, labels and R-hat history equal the oracle's."""
    import fmcmc_amd as f
    X, y = readme_data
    init = np.tile([0, 0, O.r_sd(y)], (3, 1)) + 0.2 * np.random.default_rng(4).standard_normal((3, 3))
    kw = dict(min_=-0.15, max_=0.15, lb=[-10, -10, 0.1], ub=10.0, scheme="random")
    chk = f.convergence_gelman(100, threshold=1.10)
    kern = f.kernel_unif_reflective(**kw)
    ans = f.MCMC(init, f.gaussian_linreg(X, y), 3000, seed=5, nchains=3, kernel=kern, conv_checker=chk)
    ro = O.mcmc_with_conv_checker(O.Model(O.FAM_LINREG, X, y), O.Kernel(O.K_UNIF_REFLECTIVE, 3, **kw), init, 3000, 3, 100,
                                  threshold=1.10, seed=5)
    assert [h[0] for h in chk.history] == [h[0] for h in ro.history] and len(ro.history) == 14
    assert np.array_equal(_bits(ans.as_array()), _bits(ro.samples)) and list(ans.iters) == list(ro.iters)
    assert np.array_equal(kern._state.scheme_cols.cpu().numpy()[:, 1:], ro.state.scheme_cols[:, 1:])
    assert kern.k == 1 and repr(kern).startswith("<fmcmc_kernel kernel_unif_reflective")
    # explicit sequence and kernel_ram(freq, constr) through the same front end
    a2 = f.MCMC([0, 0, O.r_sd(y)], f.gaussian_linreg(X, y), 600, seed=2, kernel=f.kernel_normal(scale=0.05, scheme=[3, 1, 2]))
    r2 = O.run(O.Model(O.FAM_LINREG, X, y), O.Kernel(O.K_NORMAL, 3, scale=0.05, scheme=[3, 1, 2]), [0, 0, O.r_sd(y)], nsteps=600, seed=2)
    assert np.array_equal(_bits(a2.data), _bits(r2.samples[0]))
    M = np.tril(np.ones((3, 3)))
    M[2, 0] = 0.0
    a3 = f.MCMC([0, 0, O.r_sd(y)], f.gaussian_linreg(X, y), 600, seed=2, kernel=f.kernel_ram(freq=2, constr=M))
    r3 = O.run(O.Model(O.FAM_LINREG, X, y), O.Kernel(O.K_RAM, 3, freq=2, constr=M), [0, 0, O.r_sd(y)], nsteps=600, seed=2)
    assert np.array_equal(_bits(a3.data), _bits(r3.samples[0]))


def test_mcmc_output_accessors(E, O, readme_data):
    """R/mcmc_info.R:301-400: get_*() return the arguments of the last MCMC() call."""
    import fmcmc_amd as f
    X, y = readme_data
    kern = f.kernel_normal(scale=0.05)
    f.MCMC([0, 0, 4.0], f.gaussian_linreg(X, y), 300, seed=9, kernel=kern, burnin=20, thin=2)
    assert f.get_nsteps() == 300 and f.get_seed() == 9 and f.get_burnin() == 20 and f.get_thin() == 2 and f.get_nchains() == 1
    assert f.get_kernel() is kern and f.get_("nsteps") == 300 and f.get_conv_checker() is None and f.get_elapsed() > 0
    assert len(f.get_("logpost")) == 140
    with pytest.raises(RuntimeError, match="not found in MCMC_OUTPUT"):
        f.get_("userdata")


def test_mcmc_api_mirror_kernels(E, O, readme_data):
    """kernel_nmirror / kernel_umirror through MCMC(): per-chain views of the adapted mean and scale like kernel[[i]]$mu in R."""
    import fmcmc_amd as f
    X, y = readme_data
    init = np.tile([3.0, 2.0, 4.0], (3, 1)) + 0.05 * np.random.default_rng(1).standard_normal((3, 3))
    for ctor, kind in ((f.kernel_nmirror, O.K_NMIRROR), (f.kernel_umirror, O.K_UMIRROR)):
        kw = dict(mu=[3.0, 2.0, 4.0], scale=0.2, warmup=300, nadapt=5, lb=[-20, -20, 0.05], ub=20.0)
        kern = ctor(**kw)
        ans = f.MCMC(init, f.gaussian_linreg(X, y), 800, seed=17, nchains=3, kernel=kern)
        ro = O.run(O.Model(O.FAM_LINREG, X, y), O.Kernel(kind, 3, **kw), init, nsteps=800, seed=17)
        assert np.array_equal(_bits(ans.as_array()), _bits(ro.samples))
        assert np.array_equal(_bits(kern[2].mu), _bits(ro.state.mirror_mu[1])) and np.array_equal(kern[3].obs_arate, ro.state.obs_arate[2], equal_nan=True)
        assert list(kern.abs_iter) == [799] * 3 and abs(ans.as_array()[:, 300:, 0].mean() - 3.1) < 0.6


def test_single_chain_checkers_in_mcmc(E, O):
    """vignettes/workflow-with-fmcmc.Rmd:95-106 (logistic model, convergence_geweke(200)) and
    inst/tinytest/test-convergence.R (convergence_auto picks Geweke for one chain, Gelman for several; Heidelberger)."""
    import fmcmc_amd as f
    rng = np.random.default_rng(7)
    n = 500
    X = rng.standard_normal((n, 2))
    yb = (rng.uniform(size=n) < 1 / (1 + np.exp(-(0.5 + X @ [1.0, -1.0])))).astype(float)
    fun = f.logistic(X, yb, intercept=True, prior_div=8.0)
    gw = f.convergence_geweke(200)
    out = f.MCMC([0.0, 0.0, 0.0], fun, 3000, seed=42, kernel=f.kernel_normal(scale=0.2), conv_checker=gw)
    assert isinstance(out, f.Mcmc) and out.niter % 200 == 0 and len(gw.history) == out.niter // 200 and "Geweke" in gw.msg
    # the GPU chain and the checker's verdicts equal a host replay on the oracle's chain
    ro = O.run(O.Model(O.FAM_LOGISTIC, X, yb, prior_div=8.0), O.Kernel(O.K_NORMAL, 3, scale=0.2), [0.0, 0.0, 0.0], nsteps=200, seed=42)
    assert np.array_equal(_bits(out.data[:200]), _bits(ro.samples[0]))
    z_host = f.geweke_diag(out.data[:200], np.arange(1, 201))
    assert np.allclose(gw.history[0][1], z_host, rtol=1e-12, equal_nan=True)
    au = f.convergence_auto(300)
    o1 = f.MCMC([0.0, 0.0, 0.0], fun, 1500, seed=1, kernel=f.kernel_normal(scale=0.2), conv_checker=au)
    assert isinstance(o1, f.Mcmc) and "Geweke" in au.msg
    o2 = f.MCMC(np.zeros((3, 3)), fun, 1500, seed=1, nchains=3, kernel=f.kernel_normal(scale=0.2), conv_checker=au)
    assert isinstance(o2, f.McmcList) and "Gelman" in au.msg
    hd = f.convergence_heildel(500)
    o3 = f.MCMC([0.0, 0.0, 0.0], fun, 2000, seed=3, kernel=f.kernel_normal(scale=0.2), conv_checker=hd)
    assert len(hd.history) >= 1 and hd.history[0][1].shape == (3, 6)
    with pytest.raises(ValueError, match="single chain"):
        f.MCMC(np.zeros((2, 3)), fun, 2000, seed=3, nchains=2, kernel=f.kernel_normal(scale=0.2), conv_checker=hd)


def test_host_pointer_entry_point_new_rows(E, O):
    """fmcmc_mcmc_run_host stages the ABI-v2 fields too: explicit / random schemes, uniform and mirror kernels,
    kernel_ram(freq, constr), kernel_adapt(bw)."""
    from fmcmc_amd import _abi as abi
    X, y = synth_linreg(1300, 2, 6)
    Cn, k, nsteps = 4, 4, 200
    init = np.ascontiguousarray(np.array([0.5, 0.5, 0.5, 4.0])[None, :] + 0.1 * np.random.default_rng(1).standard_normal((Cn, k)))
    band = (np.abs(np.subtract.outer(np.arange(k), np.arange(k))) <= 1).astype(float)
    cases = [(O.K_UNIF, dict(min_=-0.1, max_=0.1, scheme=[2, 4, 1, 3])),
             (O.K_NORMAL_REFLECTIVE, dict(scale=0.05, lb=[-9, -9, -9, 0.1], ub=9.0, scheme="random")),
             (O.K_UMIRROR, dict(mu=[0.5, 0.5, 0.5, 4.0], scale=0.1, warmup=100, nadapt=5, lb=[-9, -9, -9, 0.1], ub=9.0)),
             (O.K_RAM, dict(freq=2, constr=band)),
             (O.K_ADAPT, dict(bw=15, warmup=20, Sd=0.6))]
    P = lambda a: a.ctypes.data if a is not None else None
    for kind, okw in cases:
        ok = O.Kernel(kind, k, **okw)
        ost = O.ChainState(init, ok.kf)
        ro = O.run(O.Model(O.FAM_LINREG, X, y), ok, nsteps=nsteps, seed=5, state=ost)
        Xc = np.ascontiguousarray(X.T); yc = np.ascontiguousarray(y)
        th = init.copy(); f0 = np.zeros(Cn); abs_iter = np.zeros(Cn, np.int64)
        Sig = np.zeros((Cn, k, k)); mp = np.zeros((Cn, k)); hm = np.zeros(Cn, np.int32); ne = np.zeros(Cn, np.int32)
        cols = np.zeros((Cn, nsteps), np.int32); mmu = np.zeros((Cn, k)); msc = np.zeros((Cn, k)); oar = np.full((Cn, k), np.nan)
        samples = np.empty((Cn, k, nsteps)); lp = np.empty((Cn, nsteps)); dr = np.empty((Cn, k, nsteps))
        acc = np.zeros(Cn, np.int64); bits = np.zeros((Cn, (nsteps + 31) // 32), np.uint32)
        status = np.zeros(Cn, np.int32); sstep = np.zeros(Cn, np.int64); stheta = np.zeros((Cn, k))
        m = abi.Model(abi.FAM_GAUSSIAN_LINREG, 2, 1300, P(Xc), P(yc), 1, 1, 0.0)
        kk = abi.Kernel(kind, k, P(ok.mu), P(ok.scale), P(ok.lb), P(ok.ub), P(ok.fixed), ok.scheme, ok.freq, ok.warmup, ok.bw,
                        ok.until, ok.eps, ok.arate, ok.Sd, P(ok.scheme_seq), 0 if ok.scheme_seq is None else ok.scheme_seq.size,
                        ok.nadapt, P(ok.constr))
        r = abi.Run(Cn, nsteps, 0, 1, 5, 0, 0, 0, 0, None, None)
        st = abi.State(P(th), P(f0), P(abs_iter), P(Sig), P(mp), P(hm), P(ne), 1, 0, P(cols), P(mmu), P(msc), P(oar))
        out = abi.Out(P(samples), P(lp), P(dr), P(acc), P(bits), P(status), P(sstep), P(stheta))
        rc = abi.lib().fmcmc_mcmc_run_host(C.byref(m), C.byref(kk), C.byref(r), C.byref(st), C.byref(out), 0)
        assert rc == 0, abi.last_error()
        assert np.array_equal(_bits(samples), _bits(ro.samples_cks)), kind
        assert np.array_equal(_bits(dr), _bits(ro.draws_cks)) and np.array_equal(bits, ro.accept_bits)
        if ok.scheme == O.SCHEME_RANDOM:
            assert np.array_equal(cols[:, 1:], ost.scheme_cols[:, 1:])
        if kind == O.K_UMIRROR:
            assert np.array_equal(_bits(mmu), _bits(ost.mirror_mu)) and np.array_equal(_bits(msc), _bits(ost.mirror_scale))
            assert np.array_equal(oar, ost.obs_arate, equal_nan=True) and np.array_equal(abs_iter, ost.abs_iter)
        if kind in (O.K_RAM, O.K_ADAPT):
            assert np.array_equal(_bits(Sig), _bits(ost.Sigma))


# ---------------------------------------------------------------------------------------------------------------------
# R's own stream through the drop-in API: MCMC(..., fed = <R's draws>) on the device retraces what fmcmc prints
# ---------------------------------------------------------------------------------------------------------------------
class RStream:
    """The variates of one MCMC_without_conv_checker call in the order R draws them: chains one after the other from ONE
    Mersenne-Twister stream (serial fan-out, R/mcmc.R:643-673); per chain log(runif(nsteps)) first (R/mcmc.R:726), then the
    kernel's draws of loop steps 2..nsteps (rnorm(k'), R/kernel_normal.R:71; rt(k, k), R/kernel_ram.R:68)."""

    def __init__(self, O, seed):
        self.g = O.RRng(seed)

    def __call__(self, nchains, nsteps, kz, kernel):
        from fmcmc_amd import _abi as abi
        logu, z = np.zeros((nchains, nsteps)), np.zeros((nchains, nsteps, kz))
        for c in range(nchains):
            logu[c] = np.log(self.g.runif(nsteps))
            for i in range(1, nsteps):
                if kernel.kind != abi.KERNEL_RAM or getattr(kernel, "ram_qfun", 0) == abi.RAM_QFUN_NORMAL:
                    z[c, i] = self.g.rnorm(kz)       # (kernel_ram(qfun = function(k) rnorm(k)) included)
                else:
                    z[c, i] = self.g.rt(kz, float(kernel.ram_df) if getattr(kernel, "ram_qfun", 0) == abi.RAM_QFUN_T_DF else float(kz))
        return logu, z


@pytest.mark.filterwarnings("ignore:While using multiple chains")
def test_fed_replay_G2_G3_autostop_through_MCMC(E, O, readme_data):
    """README.md:301-339 and :370-412: two chains sharing one R stream, convergence_gelman(200): the 13 printed R-hat values
    and the stop after 2600 steps, from the device."""
    import fmcmc_amd as f
    X, y = readme_data
    init = [0, 0, O.r_sd(y)]
    chk = f.convergence_gelman(200)
    ans = f.MCMC(init, f.gaussian_linreg(X, y), 5000, nchains=2, kernel=f.kernel_normal(scale=.05), conv_checker=chk,
                 fed=RStream(O, 1215))
    assert [round(h[1], 4) for h in chk.history] == G["G2"]["rhat"]
    assert [h[0] for h in chk.history] == list(range(200, 2601, 200))
    assert ans.niter == G["G2"]["final_steps"] and ans.nchain == 2
    chk = f.convergence_gelman(200)
    kr = f.kernel_normal_reflective(scale=.05, ub=5.0, lb=[-5.0, 0.0, 0.0])
    ans = f.MCMC(init, f.gaussian_linreg(X, y, guard=False), 5000, nchains=2, kernel=kr, conv_checker=chk,
                 fed=RStream(O, 1215))
    assert [round(h[1], 4) for h in chk.history] == G["G3"]["rhat"]
    assert ans.niter == G["G3"]["final_steps"]
    a = ans.as_array()
    assert a[:, :, 0].min() >= -5 and a.max() <= 5 and a[:, :, 1:].min() >= 0


def test_fed_replay_G5_ith_step_examples_through_MCMC(E, O):
    """R/mcmc_info.R:467-543 (seed 22): the states / proposals fmcmc's roxygen examples print at steps 500..2000 and the
    running maxima of the log-posterior, from the device (sigma is the fixed second parameter)."""
    import fmcmc_amd as f
    g = O.RRng(23133)
    x = g.rnorm(200)
    y = x * 2 + g.rnorm(200)
    fun = f.gaussian_linreg(x, y, intercept=False, guard=False)
    ans = f.MCMC([0.0, 1.0], fun, 2000, kernel=f.kernel_normal(fixed=[False, True]), fed=RStream(O, 22))
    draws = f.get_draws()
    for i, (t0, t1) in G["G5"]["ith_step"].items():
        i = int(i)
        assert sig(ans.data[i - 2, 0], 7) == t0       # theta0 seen inside f at step i = ans[i-1]
        assert sig(draws[i - 1, 0], 7) == t1          # theta1 = proposal of step i
    f.MCMC([0.0, 1.0], fun, 1000, kernel=f.kernel_normal(fixed=[False, True]), fed=RStream(O, 22))
    lp = f.get_logpost()
    got = [(sig(lp[i], 7), i + 1) for i in range(1, 1000) if lp[i] > lp[:i].max()]
    assert got == [tuple(v) for v in G["G5"]["new_max"]]


def test_readme_session_through_MCMC_on_R_stream(E, O, readme_data):
    """The README's whole session (README.md:156-269) through the drop-in API on ONE R stream, as R runs it: MCMC() with the
    default kernel (G1), continued with kernel_normal(scale = .05), then kernel_ram() (G4: 1761 / 4999 accepted) and
    kernel_adapt() (G6).  G6 is a statistical target by nature (MASS::mvrnorm maps z through LAPACK eigenvectors, the engine
    through a Cholesky factor: same law, other trajectory): within 0.025 of the printed 0.5365, like the oracle."""
    import fmcmc_amd as f
    X, y = readme_data
    fun = f.gaussian_linreg(X, y)
    rs = RStream(O, 1215)
    acc = lambda a: float(np.mean(np.any(np.diff(a.data, axis=0) != 0, axis=1)))     # 1 - coda::rejectionRate
    a1 = f.MCMC([0, 0, O.r_sd(y)], fun, 5000, fed=rs)
    assert [sig(v, 4) for v in a1.data.mean(0)] == G["G1"]["mean"]
    a2 = f.MCMC(a1, fun, 5000, kernel=f.kernel_normal(scale=.05), fed=rs)
    assert round(acc(a2) * 4999) == 3641
    a3 = f.MCMC(a2, fun, 5000, kernel=f.kernel_ram(), fed=rs)
    assert sig(acc(a3), 7) == G["G4"]["ram_accept_rate"]
    a4 = f.MCMC(a2, fun, 5000, kernel=f.kernel_adapt(), fed=rs)
    assert abs(acc(a4) - G["G6"]["adapt_accept_rate"]) < 0.025, acc(a4)


@pytest.mark.parametrize("fam", ["normal", "t3", "eta"])
def test_kernel_ram_families_on_R_stream_equal_the_oracle(E, O, readme_data, fam):
    """kernel_ram(qfun = rnorm | rt(k, 3), eta = i^-0.8 k) through MCMC() on R's own stream (the draws qfun would make,
    in R's order) against the oracle restating R's loop with the same generator: the same states, bit for bit."""
    import fmcmc_amd as f
    X, y = readme_data
    kw = {"normal": dict(qfun=f.qfun_normal()), "t3": dict(qfun=f.qfun_t(3)), "eta": dict(eta=f.eta_power(0.8))}[fam]
    okw = {"normal": dict(ram_qfun=1), "t3": dict(ram_qfun=2, ram_df=3.0), "eta": dict(ram_eta_exp=0.8)}[fam]
    init = [0.0, 0.0, O.r_sd(y)]
    a = f.MCMC(init, f.gaussian_linreg(X, y), 1500, kernel=f.kernel_ram(**kw), fed=RStream(O, 77))
    g = O.RRng(77)
    ro = O.run(O.Model(O.FAM_LINREG, X, y), O.Kernel(O.K_RAM, 3, **okw), initial=np.asarray(init)[None, :], nsteps=1500,
               rng_mode=O.RNG_RMT, math_mode=O.MATH_CANON, rng=g)
    assert np.array_equal(np.asarray(a.data).view(np.uint64), np.ascontiguousarray(ro.samples[0]).view(np.uint64))
    assert 0.05 < float(np.mean(np.any(np.diff(a.data, axis=0) != 0, axis=1))) < 0.9


def test_kernel_adapt_acceptance_on_the_philox_stream(E, O, readme_data):
    """G6 again, on the engine's own stream: over six seeds the rate spreads with sd ~ 0.012 around 0.56 (the adaptation
    makes it trajectory dependent: the reference's restatement itself gives 0.522-0.556 under different eigenvector
    conventions); the printed 0.5365 must be a plausible member of that spread."""
    from fmcmc_amd import _abi as abi
    X, y = readme_data
    gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
    big = E.DBL_MAX
    start = [3.1551402264840505, 1.8035961818024204, 4.0637777462858455]   # last row of the README's second run
    gk = E.KernelSpec(abi.KERNEL_ADAPT, 3, np.zeros(3), np.ones(3), np.full(3, -big), np.full(3, big), np.zeros(3, np.uint8),
                      warmup=500, eps=1e-4)
    rates = []
    for seed in (1, 2, 3, 4, 5, 6):
        st = E.ChainState(np.asarray(start)[None, :], 3)
        r = E.sweep(gm, gk, st, 5000, seed=seed)
        rates.append(int(r.accept_count[0]) / 4999)
    rates = np.array(rates)
    assert abs(rates.mean() - G["G6"]["adapt_accept_rate"]) < 0.04 and rates.min() > 0.50 and rates.max() < 0.62, rates
    assert abs(G["G6"]["adapt_accept_rate"] - rates.mean()) < 3.5 * max(rates.std(ddof=1), 0.008), rates


def test_autostop_with_thinning_restarts_from_the_last_kept_row(E, O, readme_data):
    """R/mcmc.R:908-911: every later bulk starts from ans[niter(ans), ], the last KEPT row -- with thin = 3 and bulks of 200
    that is not the last row the loop visited.  Same bulks, history, labels and bits as the oracle's restatement."""
    import fmcmc_amd as f
    X, y = readme_data
    init = np.tile([0, 0, O.r_sd(y)], (4, 1)) + 0.3 * np.random.default_rng(3).standard_normal((4, 3))
    chk = f.convergence_gelman(200, threshold=1.02)
    ans = f.MCMC(init, f.gaussian_linreg(X, y), 3000, seed=5, nchains=4, burnin=50, thin=3, kernel=f.kernel_normal(scale=0.05),
                 conv_checker=chk)
    ro = O.mcmc_with_conv_checker(O.Model(O.FAM_LINREG, X, y), O.Kernel(O.K_NORMAL, 3, scale=0.05), init, 3000, 4, 200,
                                  threshold=1.02, burnin=50, thin=3, seed=5)
    assert len(ro.history) >= 3 and len(chk.history) == len(ro.history)      # (the checker labels a check by the last kept
    assert np.allclose([h[1] for h in chk.history], [h[1] for h in ro.history], rtol=1e-8)   # iteration, the oracle by steps run)
    assert np.array_equal(_bits(ans.as_array()), _bits(ro.samples))
    assert list(ans.iters) == list(ro.iters)
    assert len(f.get_logpost()) == 4 and f.get_logpost()[0].shape[0] == ans.niter


def test_autostop_first_bulk_without_a_kept_row_and_argument_types(E, O, readme_data):
    """burnin = 30, thin = 45, freq = 20: the first bulk (50 steps) passes the argument checks and keeps no row; the second
    bulk (20 steps < thin) is then refused with the reference's own message (R/mcmc.R:508-510 inside the second
    MCMC_without_conv_checker call) -- not with a chain error from a restart on the NaN prefill of the history.  And a
    closure / a non-kernel are refused with the documented TypeError before anything is allocated."""
    import fmcmc_amd as f
    X, y = readme_data
    init = np.tile([0, 0, O.r_sd(y)], (3, 1)) + 0.2 * np.random.default_rng(8).standard_normal((3, 3))
    with pytest.raises(ValueError, match=r"-thin- \(45\) cannot be > than -nsteps- \(20\)"):
        f.MCMC(init, f.gaussian_linreg(X, y), 400, seed=3, nchains=3, burnin=30, thin=45, kernel=f.kernel_normal(scale=0.05),
               conv_checker=f.convergence_gelman(20, threshold=0.0))
    with pytest.raises(TypeError, match="-fun- must be one of"):
        f.MCMC(init, lambda p: 0.0, 400, nchains=3, conv_checker=f.convergence_gelman(20))
    with pytest.raises(TypeError, match="-kernel- must be"):
        f.MCMC(init, f.gaussian_linreg(X, y), 400, nchains=3, kernel="normal", conv_checker=f.convergence_gelman(20))


def test_preallocated_history_equals_separate_calls(E, O):
    """fmcmc_out.ld_rows: two calls writing behind each other into one [C][k][capacity] history give the bits of two
    separately allocated calls (every kernel family of the dispatcher that the shapes below reach)."""
    import torch
    from fmcmc_amd import _abi as abi
    big = E.DBL_MAX
    for (n, p, kind, chains) in ((10000, 3, abi.KERNEL_NORMAL, 8), (10000, 3, abi.KERNEL_ADAPT, 8), (700, 2, abi.KERNEL_RAM, 5),
                                 (900, 20, abi.KERNEL_RAM, 6), (3000, 5, abi.KERNEL_NORMAL_REFLECTIVE, 4)):
        X, y = synth_linreg(n, p, 40 + p, beta=np.linspace(1.0, -1.0, p + 1))
        k = p + 2
        init = np.concatenate([np.linspace(1.0, -1.0, p + 1), [4.0]])[None, :] + 0.01 * np.random.default_rng(p).standard_normal((chains, k))
        gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
        refl = kind == abi.KERNEL_NORMAL_REFLECTIVE
        gk = E.KernelSpec(kind, k, np.zeros(k), np.full(k, 0.01), np.full(k, -6.0 if refl else -big), np.full(k, 6.0 if refl else big),
                          np.zeros(k, np.uint8), warmup=20)
        sa, sb = E.ChainState(init, k), E.ChainState(init, k)
        a1 = E.sweep(gm, gk, sa, 60, burnin=5, thin=2, seed=9)
        a2 = E.sweep(gm, gk, sa, 41, thin=2, seed=9)
        cap = a1.samples.shape[2] + a2.samples.shape[2] + 3
        f64 = dict(dtype=torch.float64, device="cuda")
        hs, hl, hd = torch.full((chains, k, cap), -7.0, **f64), torch.full((chains, cap), -7.0, **f64), torch.full((chains, k, cap), -7.0, **f64)
        b1 = E.sweep(gm, gk, sb, 60, burnin=5, thin=2, seed=9, into=(hs, hl, hd), row0=0)
        b2 = E.sweep(gm, gk, sb, 41, thin=2, seed=9, into=(hs, hl, hd), row0=b1.samples.shape[2])
        torch.cuda.synchronize()
        n1, n2 = a1.samples.shape[2], a2.samples.shape[2]
        assert np.array_equal(_bits(hs[:, :, :n1].cpu().numpy()), _bits(a1.samples.cpu().numpy())), (kind, abi.last_kernel())
        assert np.array_equal(_bits(hs[:, :, n1:n1 + n2].cpu().numpy()), _bits(a2.samples.cpu().numpy()))
        assert np.array_equal(_bits(hl[:, n1:n1 + n2].cpu().numpy()), _bits(a2.logpost.cpu().numpy()))
        assert np.array_equal(_bits(hd[:, :, :n1].cpu().numpy()), _bits(a1.draws.cpu().numpy()))
        assert (hs[:, :, n1 + n2:] == -7.0).all() and (hl[:, n1 + n2:] == -7.0).all()     # nothing written behind the rows
