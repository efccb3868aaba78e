"""The C-ABI library loads on a GPU-less machine and exports every symbol include/fmcmc_amd.h
declares; argument validation (no GPU needed) reproduces the reference's stop() messages."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def abi():
    from fmcmc_amd import _abi, build
    if build.needs_build():
        build.build()
    _abi.lib()
    return _abi


def declared_functions():
    src = open(os.path.join(ROOT, "include", "fmcmc_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fmcmc_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(abi):
    names = declared_functions()
    assert len(names) >= 11
    L = abi.lib()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert set(abi.EXPORTS) <= set(names)
    assert L.fmcmc_abi_version() == abi.ABI_VERSION == 6


def test_struct_layouts_match_the_oracle_binding(abi, O):
    """Both ctypes mirrors of the same header must agree on sizes (catches drift in either)."""
    for a, b in ((abi.Model, O.CModel), (abi.Kernel, O.CKernel), (abi.Run, O.CRun), (abi.State, O.CState),
                 (abi.Out, O.COut)):
        assert C.sizeof(a) == C.sizeof(b)
        assert [f[0] for f in a._fields_] == [f[0] for f in b._fields_]


def test_kept_rows(abi):
    L = abi.lib()
    assert L.fmcmc_kept_rows(5000, 0, 1) == 5000
    assert L.fmcmc_kept_rows(100, 20, 7) == 11      # rows 27, 34, ..., 97
    assert L.fmcmc_kept_rows(10, 0, 3) == 3


def _host_specs(abi, k=3, nsteps=100, burnin=0, thin=1, fixed=None, lb=None, ub=None, kind=1, nchains=2, p=1):
    keep = []
    def arr(a, dt=np.float64):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a.ctypes.data
    X = np.zeros((p, 10)); y = np.zeros(10)
    m = abi.Model(abi.FAM_GAUSSIAN_LINREG, p, 10, arr(X), arr(y), 1, 1, 0.0)
    kk = abi.Kernel(kind, k, arr(np.zeros(k)), arr(np.ones(k)), arr(lb if lb is not None else [-1e308] * k),
                    arr(ub if ub is not None else [1e308] * k), arr(fixed if fixed is not None else [0] * k, np.uint8),
                    0, 1, 0, 0, float("inf"), 1e-4, 0.234, 0.0)
    r = abi.Run(nchains, nsteps, burnin, thin, 1, 0, 0, 0, 0, None, None)
    return m, kk, r, keep


@pytest.mark.parametrize("kw,substr", [
    (dict(burnin=100), "-burnin- (100) cannot be >= than -nsteps- (100)."),
    (dict(thin=100), "-thin- (100) cannot be > than -nsteps- (100)."),
    (dict(thin=0), "-thin- should be >= 1."),
    (dict(nchains=0), "`nchains` must be an integer greater than 1."),
    (dict(fixed=[1, 1, 1]), "cannot be zero"),
    (dict(kind=2, lb=[0, 0, 1.0], ub=[1, 1, 1.0]), "-ub- cannot be <= than -lb-."),
    (dict(k=4), "Incorrect length of"),
])
def test_validate_reproduces_reference_messages(abi, kw, substr):
    """R/mcmc.R:501-520, R/kernel.R:9,129-132, R/kernel_normal.R:134-135."""
    m, kk, r, keep = _host_specs(abi, **kw)
    rc = abi.lib().fmcmc_validate(C.byref(m), C.byref(kk), C.byref(r))
    assert rc in (abi.ERR_ARG, abi.ERR_UNSUPPORTED)
    assert substr in abi.last_error()


def test_validate_accepts_a_good_call(abi):
    m, kk, r, keep = _host_specs(abi)
    assert abi.lib().fmcmc_validate(C.byref(m), C.byref(kk), C.byref(r)) == abi.OK


@pytest.mark.parametrize("ram,substr", [((3, 0.0, 0.0), "unknown -qfun- family"), ((2, 0.0, 0.0), "finite df > 0"),
                                        ((2, float("inf"), 0.0), "finite df > 0"), ((0, 0.0, -1.0), "exponent of -eta-"),
                                        ((0, 0.0, float("nan")), "exponent of -eta-")])
def test_validate_ram_families(abi, ram, substr):
    """fmcmc_kernel.ram_qfun / ram_df / ram_eta_exp (the built-in families behind kernel_ram's qfun and eta)."""
    m, kk, r, keep = _host_specs(abi, kind=4, lb=[-1, -1, 0.0], ub=[1, 1, 5.0])
    assert abi.lib().fmcmc_validate(C.byref(m), C.byref(kk), C.byref(r)) == abi.OK
    kk.ram_qfun, kk.ram_df, kk.ram_eta_exp = ram
    assert abi.lib().fmcmc_validate(C.byref(m), C.byref(kk), C.byref(r)) == abi.ERR_ARG
    assert substr in abi.last_error()
    kk.ram_qfun, kk.ram_df, kk.ram_eta_exp = 2, 2.5, 0.8
    assert abi.lib().fmcmc_validate(C.byref(m), C.byref(kk), C.byref(r)) == abi.OK


def numpy_gelman_partial(x, center=None):
    """Definition of the partial vector of include/fmcmc_amd.h in numpy (x: [m][N][p])."""
    m_, N, p = x.shape
    xb = x.mean(1) - (0 if center is None else center)
    Sc = np.array([np.cov(c.T, ddof=1).reshape(p, p) for c in x])
    s2 = np.array([np.diag(s) for s in Sc])
    return np.concatenate([[m_], xb.sum(0), (xb[:, :, None] * xb[:, None, :]).sum(0).ravel(), Sc.sum(0).ravel(),
                           s2.sum(0), (s2 ** 2).sum(0), (s2 * xb).sum(0), (s2 * xb ** 2).sum(0)])


@pytest.mark.parametrize("p", [1, 3, 6])
def test_gelman_finish_matches_oracle(abi, O, p):
    """Host half of convergence_gelman: partial sums -> psrf/mpsrf == the oracle's coda restatement;
    partials of two shards simply add (this is what the all-reduce relies on)."""
    rng = np.random.default_rng(p)
    m_, N = 6, 400
    x = rng.standard_normal((m_, N, p)) * (1 + rng.uniform(0, 1, (1, 1, p))) + rng.standard_normal((m_, 1, p)) * 0.4 + 3.0
    center = x[0, 0].copy()
    part = numpy_gelman_partial(x[:2], center) + numpy_gelman_partial(x[2:], center)
    assert part.size == abi.lib().fmcmc_gelman_partial_len(p)
    psrf = np.empty(p); mps = C.c_double()
    dp = C.POINTER(C.c_double)
    rc = abi.lib().fmcmc_gelman_finish(part.ctypes.data_as(dp), p, N, psrf.ctypes.data_as(dp), C.byref(mps))
    assert rc == abi.OK
    opsrf, ompsrf = O.gelman(x)
    assert np.allclose(psrf, opsrf, rtol=1e-9)
    if p > 1:
        assert abs(mps.value - ompsrf) < 1e-9 * ompsrf
    else:
        assert np.isnan(mps.value)


def test_engine_refuses_to_run_without_a_gpu(abi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from fmcmc_amd import engine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.DeviceModel(abi.FAM_GAUSSIAN_LINREG, np.zeros((4, 1)), np.zeros(4))
    assert abi.lib().fmcmc_device_count() == 0
