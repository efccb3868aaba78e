"""Transition-kernel constructors with the reference's names, arguments and error texts.

  kernel_normal            R/kernel_normal.R:26-82
  kernel_normal_reflective R/kernel_normal.R:96-177
  kernel_adapt / kernel_am R/kernel_adapt.R:54-208
  kernel_ram               R/kernel_ram.R:65-181
  kernel_unif / kernel_unif_reflective   R/kernel_unif.R:42-91, :96-170
  kernel_nmirror / kernel_umirror        R/kernel_mirror.R:3-138, :140-283
  plan_update_sequence     R/kernel.R:66-133 (scheme = "joint" | "ordered" | "random" | integer sequence)
  check_dimensions / process_bounds   R/kernel.R:3-41

A kernel object is, as in fmcmc, a mutable environment: it is expanded ("initialised") on its
first use by MCMC(), keeps per-chain state (abs_iter, Sigma, Mean_t_prev, nerrors) between calls
(R/kernel.R:218-237) and, when used with nchains > 1, behaves like an fmcmc_kernel_list:
kernel[i] is chain i's view (R/kernel.R:348-400).  The arithmetic runs in the HIP engine; this
module only carries parameters and state to and from it.
"""
import numpy as np

from . import _abi as abi

DBL_MAX = float(np.finfo(np.float64).max)


def check_dimensions(x, k, name):
    """R/kernel.R:3-17."""
    x = np.atleast_1d(np.asarray(x))
    if x.size > 1 and x.size != k:
        raise ValueError("Incorrect length of -%s-." % name)
    if x.size == 1 and k > 1:
        return np.repeat(x, k)
    return x


def process_bounds(bounds, is_lower=True):
    """R/kernel.R:25-41: NA -> +-.Machine$double.xmax."""
    if bounds is None:
        return None
    b = np.array(bounds, dtype=np.float64, copy=True)
    b[np.isnan(b)] = -DBL_MAX if is_lower else DBL_MAX
    return b


def plan_update_sequence(k, nsteps, fixed, scheme, rng=None):
    """R/kernel.R:66-133: logical [nsteps x k] plan of which parameters each step updates.  Host utility (the device derives
    the same plan on the fly); scheme = "random" draws with numpy here -- inside MCMC() the plan comes from the engine's
    counter-based stream (or from R's sample() in a fed replay)."""
    fixed = check_dimensions(fixed, k, "fixed").astype(bool)
    free = np.nonzero(~fixed)[0]
    plan = np.zeros((nsteps, k), dtype=bool)
    rows = np.arange(nsteps)
    if not isinstance(scheme, str) and np.size(scheme) > 1:
        seq = np.asarray(scheme, dtype=np.int64).reshape(-1)
        if seq.size != free.size:
            raise ValueError("When setting the update scheme, it should have the same length as the number of variables that "
                             "will not be fixed. Right now length(scheme) = %d while sum(!fixed) = %d." % (seq.size, free.size))
        missing = [int(w) + 1 for w in free if (w + 1) not in seq]
        if missing:
            raise ValueError("One or more variables was not included in the ordering sequence. The full list follows: %s. "
                             "Only variables that are not fixed can be included in this list." % ", ".join(map(str, missing)))
        plan[rows, (seq - 1)[rows % seq.size]] = True
    elif scheme == "joint":
        plan[:, free] = True
    elif scheme == "ordered":
        if free.size:
            plan[rows, free[rows % free.size]] = True
    elif scheme == "random":
        pool = np.arange(free[0] + 1) if free.size == 1 else free      # sample(x) with length-one x means 1:x in R
        if free.size:
            plan[rows, (rng or np.random.default_rng()).choice(pool, nsteps)] = True
    else:
        raise ValueError("-scheme- update must be either an integer sequence, 'joint', 'ordered', or 'random'.")
    if nsteps and plan[0].sum() == 0:
        raise ValueError("The number of parameters to update, i.e. not fixed, cannot be zero. "
                         "Check the value -fixed- in the kernel initialization.")
    return plan


class _KernelView:
    """kernel[[i]] of an fmcmc_kernel_list: one chain's state."""

    def __init__(self, parent, i):
        self._p, self._i = parent, i

    def __getattr__(self, name):
        p, i = self._p, self._i
        st = p._state
        if name == "abs_iter":
            return int(st.abs_iter[i].item()) if st is not None else 0
        if name == "Sigma":
            return st.Sigma[i].cpu().numpy() if st is not None else p.Sigma
        if name == "Mean_t_prev":
            if st is None or not int(st.have_mean[i].item()):
                return None
            return st.mean_prev[i].cpu().numpy()
        if name == "nerrors":
            return int(st.nerrors[i].item()) if st is not None else 0
        if p.kind in abi.MIRROR_KERNELS and st is not None and name in ("mu", "scale", "obs_arate"):
            t = {"mu": st.mirror_mu, "scale": st.mirror_scale, "obs_arate": st.obs_arate}[name][i]
            return t.cpu().numpy() if t.ndim else float(t.item())
        return getattr(p, name)


class fmcmc_kernel:
    def __init__(self, kind, **kw):
        self.kind = kind
        self.k = None            # set on first use, like the R closures (`k <<- length(env$theta0)`)
        self.which_ = None
        self._state = None       # engine.ChainState: per-chain persistent environments
        self._spec = None
        self._nchains = None
        for n, v in kw.items():
            setattr(self, n, v)

    # --- first-call initialisation (R/kernel_normal.R:39-63, R/kernel_adapt.R:87-115, R/kernel_ram.R:93-121)
    def _init(self, k):
        if self.k is not None and self._k_total == k:
            return
        self._k_total = k
        unif = self.kind in (abi.KERNEL_UNIF, abi.KERNEL_UNIF_REFLECTIVE)
        if unif:   # runif(k, min., max.) == min. + (max. - min.) * unif_rand(): mu := min., scale := max. - min.
            self.min_ = check_dimensions(self.min_, k, "min.").astype(np.float64)
            self.max_ = check_dimensions(self.max_, k, "max.").astype(np.float64)
            self.mu, self.scale = self.min_, self.max_ - self.min_
        else:
            self.mu = check_dimensions(self.mu, k, "mu").astype(np.float64)
            if self.kind in (abi.KERNEL_NORMAL, abi.KERNEL_NORMAL_REFLECTIVE) + abi.MIRROR_KERNELS:
                self.scale = check_dimensions(self.scale, k, "scale").astype(np.float64)
            else:
                self.scale = np.ones(k)
        if self.kind not in (abi.KERNEL_NORMAL, abi.KERNEL_UNIF):
            self.ub = process_bounds(check_dimensions(self.ub, k, "ub"), False)
            self.lb = process_bounds(check_dimensions(self.lb, k, "lb"), True)
        else:
            self.lb, self.ub = np.full(k, -DBL_MAX), np.full(k, DBL_MAX)
        self.fixed = check_dimensions(self.fixed, k, "fixed").astype(bool)
        if self.kind not in (abi.KERNEL_NORMAL, abi.KERNEL_UNIF) and np.any(self.ub <= self.lb):
            raise ValueError("-ub- cannot be <= than -lb-.")
        if unif and np.any(self.max_ <= self.min_):
            raise ValueError("-max.- cannot be <= than -min.-.")
        self.which_ = np.nonzero(~self.fixed)[0]
        kf = int(self.which_.size)
        self._scheme_seq = None
        if self.kind in abi.SIMPLE_KERNELS:
            # plan_update_sequence (R/kernel.R:66-133); explicit sequences use R's 1-based parameter positions
            if not isinstance(self.scheme, str) and np.size(self.scheme) > 1:
                seq = np.asarray(self.scheme, dtype=np.int64).reshape(-1)
                if seq.size != kf:
                    raise ValueError("When setting the update scheme, it should have the same length as the number of "
                                     "variables that will not be fixed. Right now length(scheme) = %d while "
                                     "sum(!fixed) = %d." % (seq.size, kf))
                missing = [int(w) + 1 for w in self.which_ if (w + 1) not in seq]
                if missing:
                    raise ValueError("One or more variables was not included in the ordering sequence. The full list "
                                     "follows: %s. Only variables that are not fixed can be included in this list."
                                     % ", ".join(map(str, missing)))
                self._scheme_seq = (seq - 1).astype(np.int32)
                self._scheme_id = abi.SCHEME_EXPLICIT
            elif self.scheme in ("joint", "ordered", "random"):
                self._scheme_id = {"joint": abi.SCHEME_JOINT, "ordered": abi.SCHEME_ORDERED,
                                   "random": abi.SCHEME_RANDOM}[self.scheme]
            else:
                raise ValueError("-scheme- update must be either an integer sequence, 'joint', 'ordered', or 'random'.")
        if kf == 0:
            raise ValueError("The number of parameters to update, i.e. not fixed, cannot be zero. "
                             "Check the value -fixed- in the kernel initialization.")
        if self.kind in abi.SIMPLE_KERNELS:
            self.k = kf if self._scheme_id == abi.SCHEME_JOINT else 1  # k <<- sum(update_sequence[1,])
        else:
            self.k = kf
        if self.kind == abi.KERNEL_ADAPT and self.Sd is None:
            self.Sd = 5.76 / kf
        self._kf = kf

    def spec(self, device):
        from .engine import KernelSpec
        sch = getattr(self, "_scheme_id", abi.SCHEME_JOINT) if self.kind in abi.SIMPLE_KERNELS else abi.SCHEME_JOINT
        until = getattr(self, "until", float("inf"))
        constr = getattr(self, "constr", None)
        if constr is not None:   # constr[which., , drop = FALSE][, which., drop = FALSE] (R/kernel_ram.R:150)
            constr = np.asarray(constr, dtype=np.float64)[np.ix_(self.which_, self.which_)]
        return KernelSpec(self.kind, self._k_total, self.mu, self.scale, self.lb, self.ub,
                          self.fixed.astype(np.uint8), scheme=sch, freq=getattr(self, "freq", 1),
                          warmup=getattr(self, "warmup", 0), bw=getattr(self, "bw", 0), until=until,
                          eps=getattr(self, "eps", 1e-4), arate=getattr(self, "arate", 0.234),
                          Sd=getattr(self, "Sd", 0.0) or 0.0, scheme_seq=getattr(self, "_scheme_seq", None),
                          constr=constr, nadapt=getattr(self, "nadapt", 4), ram_qfun=getattr(self, "ram_qfun", 0),
                          ram_df=getattr(self, "ram_df", 0.0), ram_eta_exp=getattr(self, "ram_eta_exp", 0.0), device=device)

    def state_for(self, initial, device):
        """Chain state for this call: the kernel's persistent part survives, theta0 := initial."""
        import torch
        from .engine import ChainState
        nch = initial.shape[0]
        if self._state is None or self._nchains != nch or self._state.device != torch.device(device):
            st = ChainState(initial, self._kf, device=device)
            Sigma0 = getattr(self, "Sigma", None)
            if self.kind in (abi.KERNEL_ADAPT, abi.KERNEL_RAM) and Sigma0 is not None:
                S0 = np.asarray(Sigma0, dtype=np.float64).reshape(self._kf, self._kf)
                st.Sigma.copy_(torch.as_tensor(np.broadcast_to(S0, (nch, self._kf, self._kf)).copy()))
                st.fresh = 0
            self._state, self._nchains = st, nch
        else:
            st = self._state
            th = torch.as_tensor(np.ascontiguousarray(initial), dtype=torch.float64) if not torch.is_tensor(initial) else initial
            st.theta0.copy_(th.to(st.device))
        return st

    def __getitem__(self, i):
        """kernel[[i]] (1-based like R when used after a multi-chain run)."""
        if self._nchains is None:
            raise IndexError("kernel has not been used with multiple chains yet")
        if not (1 <= i <= self._nchains):
            raise IndexError("chain index out of range")
        return _KernelView(self, i - 1)

    def __len__(self):
        return self._nchains or 1

    @property
    def abs_iter(self):
        if self._state is None:
            return 0
        a = self._state.abs_iter.cpu().numpy()
        return int(a[0]) if a.size == 1 else a

    def __repr__(self):
        names = {1: "kernel_normal", 2: "kernel_normal_reflective", 3: "kernel_adapt", 4: "kernel_ram",
                 5: "kernel_unif", 6: "kernel_unif_reflective", 7: "kernel_nmirror", 8: "kernel_umirror"}
        return "<fmcmc_kernel %s k=%s>" % (names[self.kind], self.k)


def kernel_normal(mu=0.0, scale=1.0, fixed=False, scheme="joint"):
    return fmcmc_kernel(abi.KERNEL_NORMAL, mu=mu, scale=scale, fixed=fixed, scheme=scheme)


def kernel_normal_reflective(mu=0.0, scale=1.0, lb=-DBL_MAX, ub=DBL_MAX, fixed=False, scheme="joint"):
    return fmcmc_kernel(abi.KERNEL_NORMAL_REFLECTIVE, mu=mu, scale=scale, lb=lb, ub=ub, fixed=fixed,
                        scheme=scheme)


def kernel_unif(min_=-1.0, max_=1.0, fixed=False, scheme="joint"):
    """R/kernel_unif.R:42-91 (`min.` / `max.` are spelled min_ / max_)."""
    return fmcmc_kernel(abi.KERNEL_UNIF, min_=min_, max_=max_, mu=0.0, fixed=fixed, scheme=scheme)


def kernel_unif_reflective(min_=-1.0, max_=1.0, lb=None, ub=None, fixed=False, scheme="joint"):
    """R/kernel_unif.R:96-170; lb / ub default to min. / max. (:99-100)."""
    return fmcmc_kernel(abi.KERNEL_UNIF_REFLECTIVE, min_=min_, max_=max_, mu=0.0, lb=min_ if lb is None else lb,
                        ub=max_ if ub is None else ub, fixed=fixed, scheme=scheme)


def kernel_nmirror(mu=0.0, scale=1.0, warmup=500, nadapt=4, arate=0.4, lb=-DBL_MAX, ub=DBL_MAX, fixed=False, scheme="joint"):
    """R/kernel_mirror.R:3-138: theta1 ~ N(2 mu - theta0, scale^2), mu = running mean during warm-up, scale rescaled once at
    abs_iter == nadapt by tan(pi/2 obs_arate) / tan(pi/2 arate)."""
    return fmcmc_kernel(abi.KERNEL_NMIRROR, mu=mu, scale=scale, warmup=int(warmup), nadapt=int(nadapt), arate=arate, lb=lb, ub=ub,
                        fixed=fixed, scheme=scheme)


def kernel_umirror(mu=0.0, scale=1.0, warmup=500, nadapt=4, arate=0.4, lb=-DBL_MAX, ub=DBL_MAX, fixed=False, scheme="joint"):
    """R/kernel_mirror.R:140-283: theta1 ~ U(2 mu - theta0 -+ sqrt(3) scale), same adaptation."""
    return fmcmc_kernel(abi.KERNEL_UMIRROR, mu=mu, scale=scale, warmup=int(warmup), nadapt=int(nadapt), arate=arate, lb=lb, ub=ub,
                        fixed=fixed, scheme=scheme)


def kernel_adapt(mu=0.0, bw=0, lb=-DBL_MAX, ub=DBL_MAX, freq=1, warmup=500, Sigma=None, Sd=None,
                 eps=1e-4, fixed=False, until=float("inf")):
    if bw > 0 and bw > warmup:
        raise ValueError("The `warmup` parameter must be greater than `bw`.")
    if int(freq) < 1 or int(bw) < 0:
        raise ValueError("-freq- must be >= 1 and -bw- >= 0.")
    return fmcmc_kernel(abi.KERNEL_ADAPT, mu=mu, bw=int(bw), lb=lb, ub=ub, freq=int(freq), warmup=int(warmup),
                        Sigma=Sigma, Sd=Sd, eps=eps, fixed=fixed, until=until)


kernel_am = kernel_adapt


class eta_power:
    """kernel_ram(eta = function(i, k) min(c(1.0, i^(-exponent) * k))): the family the default belongs to
    (exponent = 2/3, R/kernel_ram.R:67); any exponent in (0, 1] keeps the adaptation diminishing (Vihola 2012)."""

    def __init__(self, exponent=2.0 / 3.0):
        if not (0.0 < float(exponent) < float("inf")):
            raise ValueError("the exponent of -eta- must be finite and positive.")
        self.exponent = float(exponent)

    def __call__(self, i, k):
        return min(1.0, float(i) ** (-self.exponent) * k)


class qfun_t:
    """kernel_ram(qfun = function(k) stats::rt(k, df)); df = None is the default rt(k, k) (R/kernel_ram.R:68)."""

    def __init__(self, df=None):
        if df is not None and not (0.0 < float(df) < float("inf")):
            raise ValueError("-df- must be finite and positive.")
        self.df = None if df is None else float(df)


class qfun_normal:
    """kernel_ram(qfun = function(k) stats::rnorm(k)): the Gaussian variates of Vihola (2012)."""


def kernel_ram(mu=0.0, eta=None, qfun=None, arate=0.234, freq=1, warmup=0, Sigma=None, eps=1e-4,
               lb=-DBL_MAX, ub=DBL_MAX, fixed=False, until=float("inf"), constr=None):
    """R/kernel_ram.R:65-176.  eta / qfun are R closures there; closures cannot run inside the device loop, so the engine
    offers the families they are used for: eta = eta_power(exponent), qfun = qfun_t(df) | qfun_normal() (None = the
    reference's defaults).  Anything else raises."""
    ram_qfun, ram_df, ram_eta_exp = abi.RAM_QFUN_T_K, 0.0, 0.0
    if isinstance(qfun, qfun_normal) or qfun == "normal":
        ram_qfun = abi.RAM_QFUN_NORMAL
    elif isinstance(qfun, qfun_t):
        if qfun.df is not None:
            ram_qfun, ram_df = abi.RAM_QFUN_T_DF, qfun.df
    elif qfun is not None:
        raise NotImplementedError("-qfun- must be None (rt(k, k), R/kernel_ram.R:68), qfun_t(df) or qfun_normal(): an "
                                  "arbitrary closure cannot run inside the device loop.")
    if isinstance(eta, eta_power):
        ram_eta_exp = eta.exponent
    elif eta is not None:
        raise NotImplementedError("-eta- must be None (min(1, k i^(-2/3)), R/kernel_ram.R:67) or eta_power(exponent): an "
                                  "arbitrary closure cannot run inside the device loop.")
    if int(freq) < 1:
        raise ValueError("-freq- must be >= 1.")
    return fmcmc_kernel(abi.KERNEL_RAM, mu=mu, arate=arate, freq=int(freq), warmup=int(warmup), Sigma=Sigma,
                        eps=eps, lb=lb, ub=ub, fixed=fixed, until=until, constr=constr, eta=eta, qfun=qfun,
                        ram_qfun=ram_qfun, ram_df=ram_df, ram_eta_exp=ram_eta_exp)
