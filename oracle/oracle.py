"""ctypes front end of the CPU oracle -- TEST INFRASTRUCTURE, never the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
It wraps oracle/build/libfmcmc_oracle.so (built from fmcmc_oracle.c + r_rng.c by
oracle/Makefile) and restates, in Python, the host-side orchestration of the reference that
sits around the per-chain loop:

  R/mcmc.R:841-1019   MCMC_with_conv_checker (bulks, restart from last row)  -> mcmc_with_conv_checker
  R/append_chains.R:90-143  iteration labels across bulks                      -> inside the same
  R/convergence.R:191-246   convergence_gelman (window = second half, mpsrf)   -> gelman_check
  R/checks.R:22-58    check_initial (recycling)                                -> _as_initial
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "libfmcmc_oracle.so")

RNG_PHILOX, RNG_RMT = 0, 1
MATH_CANON, MATH_R = 0, 1

FAM_LINREG, FAM_LOGISTIC, FAM_IID_NORMAL = 1, 2, 3
K_NORMAL, K_NORMAL_REFLECTIVE, K_ADAPT, K_RAM, K_UNIF, K_UNIF_REFLECTIVE, K_NMIRROR, K_UMIRROR = 1, 2, 3, 4, 5, 6, 7, 8
SCHEME_JOINT, SCHEME_ORDERED, SCHEME_RANDOM, SCHEME_EXPLICIT = 0, 1, 2, 3
DBL_MAX = np.finfo(np.float64).max

_dp = C.POINTER(C.c_double)


class CModel(C.Structure):
    _fields_ = [("family", C.c_int32), ("p", C.c_int32), ("n", C.c_int64), ("X", _dp), ("y", _dp),
                ("intercept", C.c_int32), ("guard", C.c_int32), ("prior_div", C.c_double)]


class CKernel(C.Structure):
    _fields_ = [("kind", C.c_int32), ("k", C.c_int32), ("mu", _dp), ("scale", _dp), ("lb", _dp),
                ("ub", _dp), ("fixed", C.POINTER(C.c_uint8)), ("scheme", C.c_int32),
                ("freq", C.c_int32), ("warmup", C.c_int32), ("bw", C.c_int32), ("until", C.c_double),
                ("eps", C.c_double), ("arate", C.c_double), ("Sd", C.c_double),
                ("scheme_seq", C.POINTER(C.c_int32)), ("scheme_len", C.c_int32), ("nadapt", C.c_int32),
                ("constr", _dp), ("h_fixed", C.c_void_p), ("h_lb", C.c_void_p), ("h_ub", C.c_void_p),
                ("h_scale", C.c_void_p), ("h_scheme_seq", C.c_void_p),   # (v3 host mirrors: device entry only)
                ("ram_qfun", C.c_int32), ("reserved", C.c_int32), ("ram_df", C.c_double), ("ram_eta_exp", C.c_double)]


class CRun(C.Structure):
    _fields_ = [("nchains", C.c_int64), ("nsteps", C.c_int64), ("burnin", C.c_int64),
                ("thin", C.c_int64), ("seed", C.c_uint64), ("chain_base", C.c_int64),
                ("step_base", C.c_int64), ("rng_mode", C.c_int32), ("reserved", C.c_int32),
                ("fed_logu", _dp), ("fed_z", _dp)]


class CState(C.Structure):
    _fields_ = [("theta0", _dp), ("f0", _dp), ("abs_iter", C.POINTER(C.c_int64)), ("Sigma", _dp),
                ("mean_prev", _dp), ("have_mean", C.POINTER(C.c_int32)),
                ("nerrors", C.POINTER(C.c_int32)), ("fresh", C.c_int32), ("reserved", C.c_int32),
                ("scheme_cols", C.POINTER(C.c_int32)), ("mirror_mu", _dp), ("mirror_scale", _dp), ("obs_arate", _dp)]


class COut(C.Structure):
    _fields_ = [("samples", _dp), ("logpost", _dp), ("draws", _dp),
                ("accept_count", C.POINTER(C.c_int64)), ("accept_bits", C.POINTER(C.c_uint32)),
                ("status", C.POINTER(C.c_int32)), ("status_step", C.POINTER(C.c_int64)),
                ("status_theta", _dp), ("ld_rows", C.c_int64)]   # (ld_rows: device entry only, the oracle's rows are dense)


def build(force=False):
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
            for f in ("fmcmc_oracle.c", "r_rng.c", "r_rng.h", "../include/fmh_detmath.h",
                      "../include/fmh_philox.h", "../include/fmcmc_amd.h")
            if os.path.exists(os.path.join(_HERE, f))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.r_rng_new.restype = C.c_void_p
        L.r_rng_free.argtypes = [C.c_void_p]
        L.r_rng_count.restype = C.c_uint64
        L.r_rng_count.argtypes = [C.c_void_p]
        L.r_set_seed.argtypes = [C.c_void_p, C.c_uint32]
        for nm in ("r_unif_rand", "r_norm_rand", "r_exp_rand"):
            getattr(L, nm).restype = C.c_double
            getattr(L, nm).argtypes = [C.c_void_p]
        L.r_rt.restype = C.c_double
        L.r_rt.argtypes = [C.c_void_p, C.c_double]
        L.r_rgamma.restype = C.c_double
        L.r_rgamma.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.r_qnorm_std.restype = C.c_double
        L.r_qnorm_std.argtypes = [C.c_double]
        L.r_runif_vec.argtypes = [C.c_void_p, C.c_int64, _dp]
        L.r_rnorm_vec.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_double, _dp]
        L.r_rt_vec.argtypes = [C.c_void_p, C.c_int64, C.c_double, _dp]
        L.fmcmc_oracle_run.restype = C.c_int
        L.fmcmc_oracle_run.argtypes = [C.POINTER(CModel), C.POINTER(CKernel), C.POINTER(CRun),
                                       C.POINTER(CState), C.POINTER(COut), C.c_int, C.c_int,
                                       C.c_void_p]
        L.fmcmc_oracle_logpost.restype = C.c_double
        L.fmcmc_oracle_logpost.argtypes = [C.POINTER(CModel), _dp, C.c_int]
        L.fmcmc_oracle_reflect.argtypes = [_dp, _dp, _dp, C.POINTER(C.c_int32), C.c_int, C.c_int]
        L.fmcmc_oracle_mean_recursive.argtypes = [_dp, _dp, C.c_double, C.c_int, _dp]
        L.fmcmc_oracle_cov_recursive.argtypes = [_dp, _dp, _dp, _dp, C.c_double, C.c_double,
                                                 C.c_double, _dp, C.c_int]
        L.fmcmc_oracle_gelman.restype = C.c_int
        L.fmcmc_oracle_gelman.argtypes = [_dp, C.c_int64, C.c_int, C.c_int64, _dp, _dp]
        L.fmcmc_oracle_detmath.argtypes = [C.c_int, _dp, _dp, C.c_int64]
        L.fmcmc_oracle_detmath_rng.argtypes = [C.c_int, _dp, _dp, C.c_int64, C.c_uint64]
        L.fmcmc_oracle_philox.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]
        L.fmcmc_oracle_canon_draw.restype = C.c_double
        L.fmcmc_oracle_canon_draw.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_uint32,
                                              C.c_uint32, C.c_double]
        L.fmcmc_oracle_r_sample_replace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32)]
        L.fmcmc_oracle_cpu_has_fma.restype = C.c_int
        if not L.fmcmc_oracle_cpu_has_fma():
            raise RuntimeError("oracle needs a CPU with FMA (canonical math uses fma())")
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


# ---------------------------------------------------------------------------- R's RNG
class RRng:
    """R's default generator (Mersenne-Twister / inversion); set.seed + draws."""

    def __init__(self, seed=None):
        self._h = C.c_void_p(lib().r_rng_new())
        if seed is not None:
            self.set_seed(seed)

    def __del__(self):
        try:
            lib().r_rng_free(self._h)
        except Exception:
            pass

    def set_seed(self, seed):
        lib().r_set_seed(self._h, C.c_uint32(int(seed) & 0xFFFFFFFF))

    @property
    def count(self):
        return int(lib().r_rng_count(self._h))

    def runif(self, n):
        out = np.empty(n)
        lib().r_runif_vec(self._h, n, _p(out))
        return out

    def rnorm(self, n, mean=0.0, sd=1.0):
        out = np.empty(n)
        lib().r_rnorm_vec(self._h, n, float(mean), float(sd), _p(out))
        return out

    def sample_int(self, n, size):
        """sample.int(n, size, replace = TRUE) (1-based, R >= 3.6 rejection sampling)."""
        out = np.empty(size, dtype=np.int32)
        lib().fmcmc_oracle_r_sample_replace(self._h, int(n), int(size), out.ctypes.data_as(C.POINTER(C.c_int32)))
        return out + 1

    def rt(self, n, df):
        out = np.empty(n)
        lib().r_rt_vec(self._h, n, float(df), _p(out))
        return out


def r_sd(x):
    """stats::sd: two-pass with long-double accumulation (R's cov.c)."""
    x = np.asarray(x, dtype=np.longdouble)
    n = x.size
    m = x.sum() / n
    m = m + (x - m).sum() / n
    return float(np.sqrt(((x - m) ** 2).sum() / (n - 1)))


# ---------------------------------------------------------------------------- specs
class Model:
    def __init__(self, family, X, y, intercept=True, guard=True, prior_div=0.0):
        y = _f64(y)
        n = y.shape[0]
        if X is None:
            Xc = np.zeros((0, n))
        else:
            X = np.asarray(X, dtype=np.float64)
            if X.ndim == 1:
                X = X[:, None]
            Xc = np.ascontiguousarray(X.T)  # [p][n] == column-major n x p
        self.family, self.y, self.Xc = family, y, Xc
        self.n, self.p = n, Xc.shape[0]
        self.intercept, self.guard, self.prior_div = int(bool(intercept)), int(bool(guard)), float(prior_div)

    @property
    def k(self):
        if self.family == FAM_LINREG:
            return self.intercept + self.p + 1
        if self.family == FAM_LOGISTIC:
            return self.intercept + self.p
        return 2

    def c(self):
        m = CModel(self.family, self.p, self.n, _p(self.Xc) if self.p else None, _p(self.y),
                   self.intercept, self.guard, self.prior_div)
        m._keep = (self.Xc, self.y)
        return m

    def logpost(self, theta, math_mode=MATH_CANON):
        th = _f64(theta)
        cm = self.c()
        return float(lib().fmcmc_oracle_logpost(C.byref(cm), _p(th), math_mode))


def _rec(x, k, name):
    """check_dimensions (R/kernel.R:3-17)."""
    x = np.atleast_1d(np.asarray(x))
    if x.size > 1 and x.size != k:
        raise ValueError("Incorrect length of -%s-." % name)
    if x.size == 1 and k > 1:
        x = np.repeat(x, k)
    return x


class Kernel:
    def __init__(self, kind, k, mu=0.0, scale=1.0, lb=-DBL_MAX, ub=DBL_MAX, fixed=False,
                 scheme="joint", freq=1, warmup=None, bw=0, until=np.inf, eps=1e-4, arate=0.234,
                 Sd=None, constr=None, min_=None, max_=None, nadapt=4, ram_qfun=0, ram_df=0.0, ram_eta_exp=0.0):
        self.kind, self.k = kind, k
        if kind in (K_UNIF, K_UNIF_REFLECTIVE):   # R/kernel_unif.R: runif(k, min., max.) = min. + (max. - min.) * u
            mn = _f64(_rec(-1.0 if min_ is None else min_, k, "min."))
            mx = _f64(_rec(1.0 if max_ is None else max_, k, "max."))
            if kind == K_UNIF_REFLECTIVE and np.isscalar(lb) and lb == -DBL_MAX and np.isscalar(ub) and ub == DBL_MAX:
                lb, ub = mn, mx                       # defaults lb = min., ub = max. (R/kernel_unif.R:99-100)
            if np.any(mx <= mn):
                raise ValueError("-max.- cannot be <= than -min.-.")
            mu, scale = mn, mx - mn
        self.mu = _f64(_rec(mu, k, "mu"))
        self.scale = _f64(_rec(scale, k, "scale"))
        lb = _f64(_rec(lb, k, "lb")).copy()
        ub = _f64(_rec(ub, k, "ub")).copy()
        lb[np.isnan(lb)] = -DBL_MAX  # process_bounds (R/kernel.R:25-41)
        ub[np.isnan(ub)] = DBL_MAX
        self.lb, self.ub = lb, ub
        self.fixed = np.ascontiguousarray(_rec(fixed, k, "fixed").astype(np.uint8))
        self.scheme_seq = None
        if not isinstance(scheme, str):               # explicit sequence of 1-based parameter positions (R/kernel.R:69-92)
            seq = np.asarray(scheme, dtype=np.int64)
            free = np.nonzero(self.fixed == 0)[0] + 1
            if seq.size != free.size:
                raise ValueError("When setting the update scheme, it should have the same length as the number of "
                                 "variables that will not be fixed.")
            if not np.all(np.isin(free, seq)):
                raise ValueError("One or more variables was not included in the ordering sequence.")
            self.scheme_seq = np.ascontiguousarray(seq - 1, dtype=np.int32)
            self.scheme = SCHEME_EXPLICIT
        else:
            self.scheme = {"joint": SCHEME_JOINT, "ordered": SCHEME_ORDERED, "random": SCHEME_RANDOM}[scheme]
        self.freq, self.bw, self.nadapt = int(freq), int(bw), int(nadapt)
        self.ram_qfun, self.ram_df, self.ram_eta_exp = int(ram_qfun), float(ram_df), float(ram_eta_exp)
        if warmup is None:
            warmup = 500 if kind in (K_ADAPT, K_NMIRROR, K_UMIRROR) else 0
        if kind in (K_NMIRROR, K_UMIRROR) and arate == 0.234:
            arate = 0.4                                   # default of the mirror kernels (R/kernel_mirror.R:8)
        self.warmup, self.until, self.eps, self.arate = int(warmup), float(until), float(eps), float(arate)
        self.kf = int((self.fixed == 0).sum())
        self.Sd = float(Sd) if Sd is not None else 5.76 / max(self.kf, 1)
        if kind not in (K_NORMAL, K_UNIF) and np.any(self.ub <= self.lb):
            raise ValueError("-ub- cannot be <= than -lb-.")
        free = np.nonzero(self.fixed == 0)[0]
        self.constr = None if constr is None else _f64(np.asarray(constr, dtype=np.float64)[np.ix_(free, free)])
        if self.kf == 0:
            raise ValueError("The number of parameters to update, i.e. not fixed, cannot be zero.")

    def c(self):
        kk = CKernel(self.kind, self.k, _p(self.mu), _p(self.scale), _p(self.lb), _p(self.ub),
                     self.fixed.ctypes.data_as(C.POINTER(C.c_uint8)), self.scheme, self.freq,
                     self.warmup, self.bw, self.until, self.eps, self.arate, self.Sd,
                     self.scheme_seq.ctypes.data_as(C.POINTER(C.c_int32)) if self.scheme_seq is not None else None,
                     0 if self.scheme_seq is None else int(self.scheme_seq.size), self.nadapt,
                     _p(self.constr) if self.constr is not None else None, None, None, None, None, None,
                     self.ram_qfun, 0, self.ram_df, self.ram_eta_exp)
        return kk


class ChainState:
    """Per-chain state that survives between calls (kernel envs + last row)."""

    def __init__(self, initial, kf):
        self.theta0 = _f64(initial).copy()
        Cn, k = self.theta0.shape
        self.f0 = np.zeros(Cn)
        self.abs_iter = np.zeros(Cn, dtype=np.int64)
        self.Sigma = np.zeros((Cn, kf, kf))
        self.mean_prev = np.zeros((Cn, kf))
        self.have_mean = np.zeros(Cn, dtype=np.int32)
        self.nerrors = np.zeros(Cn, dtype=np.int32)
        self.scheme_cols = None   # [C][nsteps] int32 plan of scheme = "random" (set by run())
        self.mirror_mu, self.mirror_scale = np.zeros((Cn, k)), np.zeros((Cn, k))
        self.obs_arate = np.full((Cn, k), np.nan)
        self.fresh = 1
        self.step_base = 0

    def c(self):
        return CState(_p(self.theta0), _p(self.f0), self.abs_iter.ctypes.data_as(C.POINTER(C.c_int64)),
                      _p(self.Sigma), _p(self.mean_prev),
                      self.have_mean.ctypes.data_as(C.POINTER(C.c_int32)),
                      self.nerrors.ctypes.data_as(C.POINTER(C.c_int32)), self.fresh, 0,
                      self.scheme_cols.ctypes.data_as(C.POINTER(C.c_int32)) if self.scheme_cols is not None else None,
                      _p(self.mirror_mu), _p(self.mirror_scale), _p(self.obs_arate))


def _as_initial(initial, nchains):
    """check_initial (R/checks.R:22-58)."""
    a = np.asarray(initial, dtype=np.float64)
    if a.ndim == 1:
        if a.size == 0:
            raise ValueError("The `initial` vector is of length zero.")
        a = np.tile(a, (nchains, 1))
    elif a.shape[0] != nchains:
        raise ValueError("The number of rows of `initial` (%d) must coincide with the number of chains (%d)."
                         % (a.shape[0], nchains))
    return np.ascontiguousarray(a)


class Result:
    pass


def run(model, kernel, initial=None, nsteps=1000, burnin=0, thin=1, seed=0, nchains=None,
        chain_base=0, rng_mode=RNG_PHILOX, math_mode=MATH_CANON, rng=None, state=None,
        want_draws=True):
    """One MCMC_without_conv_checker call (R/mcmc.R:485-838) over all chains."""
    L = lib()
    if burnin >= nsteps:
        raise ValueError("-burnin- (%d) cannot be >= than -nsteps- (%d)." % (burnin, nsteps))
    if thin >= nsteps:
        raise ValueError("-thin- (%d) cannot be > than -nsteps- (%d)." % (thin, nsteps))
    if thin < 1:
        raise ValueError("-thin- should be >= 1.")
    if state is None:
        if nchains is None:
            nchains = 1 if np.asarray(initial).ndim == 1 else np.asarray(initial).shape[0]
        state = ChainState(_as_initial(initial, nchains), kernel.kf)
    Cn, k = state.theta0.shape
    S = (nsteps - burnin) // thin
    nwords = (nsteps + 31) // 32
    samples = np.empty((Cn, k, S))
    logpost = np.empty((Cn, S))
    draws = np.empty((Cn, k, S)) if want_draws else None
    acc = np.zeros(Cn, dtype=np.int64)
    bits = np.zeros((Cn, nwords), dtype=np.uint32)
    status = np.zeros(Cn, dtype=np.int32)
    sstep = np.zeros(Cn, dtype=np.int64)
    stheta = np.zeros((Cn, k))
    out = COut(_p(samples), _p(logpost), _p(draws) if want_draws else None,
               acc.ctypes.data_as(C.POINTER(C.c_int64)), bits.ctypes.data_as(C.POINTER(C.c_uint32)),
               status.ctypes.data_as(C.POINTER(C.c_int32)), sstep.ctypes.data_as(C.POINTER(C.c_int64)),
               _p(stheta))
    crun = CRun(Cn, nsteps, burnin, thin, seed, chain_base, state.step_base, 0, 0, None, None)
    if kernel.scheme == SCHEME_RANDOM:
        if state.scheme_cols is None:
            state.scheme_cols = np.zeros((Cn, nsteps), dtype=np.int32)
        elif state.scheme_cols.shape[1] < nsteps:
            raise IndexError("subscript out of bounds")   # the plan has the rows of the kernel's first call only
    cm, ck = model.c(), kernel.c()
    full_plan = state.scheme_cols
    if full_plan is not None and full_plan.shape[1] > nsteps:    # a shorter call (last bulk) reads the first rows of the plan
        state.scheme_cols = np.ascontiguousarray(full_plan[:, :nsteps])
    cs = state.c()
    state.scheme_cols, cs._keep = full_plan, state.scheme_cols
    if rng_mode == RNG_RMT and rng is None:
        raise ValueError("RMT mode needs an RRng")
    rc = L.fmcmc_oracle_run(C.byref(cm), C.byref(ck), C.byref(crun), C.byref(cs), C.byref(out),
                            rng_mode, math_mode, rng._h if rng is not None else None)
    if rc not in (0, 3):
        raise RuntimeError("oracle run failed rc=%d" % rc)
    state.fresh = 0
    state.step_base += nsteps
    r = Result()
    r.rc = rc
    r.samples = samples.transpose(0, 2, 1).copy()  # [C][S][k]
    r.samples_cks = samples                         # [C][k][S] ABI layout
    r.logpost = logpost
    r.draws = draws.transpose(0, 2, 1).copy() if want_draws else None
    r.draws_cks = draws
    r.accept_count, r.accept_bits = acc, bits
    r.status, r.status_step, r.status_theta = status, sstep, stheta
    r.state = state
    r.iters = burnin + thin * np.arange(1, S + 1)  # row labels (R/mcmc.R:728-729,815-816)
    r.thin = thin
    return r


def accept_steps(bits_row):
    """Loop indices i (R's 1-based) at which the proposal was accepted."""
    idx = []
    for w, word in enumerate(bits_row):
        word = int(word)
        while word:
            b = (word & -word).bit_length() - 1
            idx.append(w * 32 + b + 1)
            word &= word - 1
    return np.array(idx, dtype=np.int64)


# ---------------------------------------------------------------------------- Gelman-Rubin
def gelman(chains):
    """coda::gelman.diag point estimates. chains: [m][N][p]. Returns (psrf[p], mpsrf)."""
    x = np.ascontiguousarray(np.asarray(chains, dtype=np.float64).transpose(0, 2, 1))  # [m][p][N]
    m, p, N = x.shape
    psrf = np.empty(p)
    mps = C.c_double()
    rc = lib().fmcmc_oracle_gelman(_p(x), m, p, N, _p(psrf), C.byref(mps))
    if rc == 1:
        raise ValueError("gelman needs >= 2 chains and >= 2 iterations")
    return psrf, mps.value


def gelman_check(chains, iters, threshold=1.10):
    """convergence_gelman closure (R/convergence.R:198-244) incl. coda's autoburnin window
    (keep iterations > end/2 when start < end/2) and rm_invariant's whole-block test."""
    chains = np.asarray(chains)
    iters = np.asarray(iters)
    allv = chains.reshape(-1)
    if allv.size > 1 and np.var(allv, ddof=1) < 1e-10:  # rm_invariant quirk: sd of ALL entries
        return False, float("nan")
    start, end = iters[0], iters[-1]
    if start < end / 2:
        keep = iters >= end / 2 + 1
        chains = chains[:, keep, :]
    try:
        psrf, mpsrf = gelman(chains)
    except ValueError:
        return False, float("nan")
    p = chains.shape[2]
    val = mpsrf if p > 1 else psrf[0]
    if not np.isfinite(val):
        return False, val
    return bool(val < threshold), float(val)


def mcmc_with_conv_checker(model, kernel, initial, nsteps, nchains, freq, threshold=1.10, burnin=0,
                           thin=1, seed=0, rng_mode=RNG_PHILOX, math_mode=MATH_CANON, rng=None):
    """MCMC_with_conv_checker (R/mcmc.R:841-1019) with convergence_gelman(freq, threshold)."""
    if freq * 2 > nsteps:
        freq = 0
    if freq > 0:
        bulks = [freq] * ((nsteps - burnin) // freq)
        if (nsteps - burnin) % freq:
            bulks.append((nsteps - burnin) - sum(bulks))
    else:
        bulks = [nsteps]
    bulks[0] += burnin
    state = ChainState(_as_initial(initial, nchains), kernel.kf)
    free = np.where(kernel.fixed == 0)[0]
    parts, iters, history = [], None, []
    converged = False
    for bi, nb in enumerate(bulks):
        r = run(model, kernel, nsteps=nb, burnin=burnin if bi == 0 else 0, thin=thin, seed=seed,
                rng_mode=rng_mode, math_mode=math_mode, rng=rng, state=state)
        parts.append(r.samples)
        # initial <- ans[niter(ans), ]: the next bulk starts from the last KEPT row (R/mcmc.R:908-911); with thin > 1 that
        # is not the last row the loop visited
        state.theta0[:] = r.samples[:, -1, :]
        if iters is None:
            iters = r.iters.copy()
        else:  # append_chains.mcmc labels (R/append_chains.R:113-121)
            iters = np.concatenate([iters, r.iters - r.iters[0] + iters[-1] + thin])
        ans = np.concatenate(parts, axis=1)
        converged, val = gelman_check(ans[:, :, free], iters, threshold)
        history.append((int(sum(bulks[:bi + 1])), val))
        if converged:
            break
    res = Result()
    res.samples, res.iters, res.history, res.converged, res.state = ans, iters, history, converged, state
    return res
