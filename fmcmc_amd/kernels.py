"""Transition-kernel constructors with the reference's names, arguments and error texts.

  kernel_normal            R/kernel_normal.R:26-82
  kernel_normal_reflective R/kernel_normal.R:96-177
  kernel_adapt / kernel_am R/kernel_adapt.R:54-208
  kernel_ram               R/kernel_ram.R:65-181
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
        self.mu = check_dimensions(self.mu, k, "mu").astype(np.float64)
        if self.kind in (abi.KERNEL_NORMAL, abi.KERNEL_NORMAL_REFLECTIVE):
            self.scale = check_dimensions(self.scale, k, "scale").astype(np.float64)
        else:
            self.scale = np.ones(k)
        if self.kind != abi.KERNEL_NORMAL:
            self.ub = process_bounds(check_dimensions(self.ub, k, "ub"), False)
            self.lb = process_bounds(check_dimensions(self.lb, k, "lb"), True)
        else:
            self.lb, self.ub = np.full(k, -DBL_MAX), np.full(k, DBL_MAX)
        self.fixed = check_dimensions(self.fixed, k, "fixed").astype(bool)
        if self.kind != abi.KERNEL_NORMAL and np.any(self.ub <= self.lb):
            raise ValueError("-ub- cannot be <= than -lb-.")
        self.which_ = np.nonzero(~self.fixed)[0]
        kf = int(self.which_.size)
        if kf == 0:
            raise ValueError("The number of parameters to update, i.e. not fixed, cannot be zero. "
                             "Check the value -fixed- in the kernel initialization.")
        if self.kind in (abi.KERNEL_NORMAL, abi.KERNEL_NORMAL_REFLECTIVE):
            if self.scheme not in ("joint", "ordered"):
                if self.scheme == "random" or not isinstance(self.scheme, str):
                    raise NotImplementedError("scheme = 'random' / explicit sequences draw from R's RNG and are "
                                              "not available on the device (SURVEY.md section 8(f) rank 3).")
                raise ValueError("-scheme- update must be either an integer sequence, 'joint', 'ordered', or 'random'.")
            self.k = kf if self.scheme == "joint" else 1  # k <<- sum(update_sequence[1,])
        else:
            self.k = kf
        if self.kind == abi.KERNEL_ADAPT and self.Sd is None:
            self.Sd = 5.76 / kf
        self._kf = kf

    def spec(self, device):
        from .engine import KernelSpec
        sch = abi.SCHEME_ORDERED if getattr(self, "scheme", "joint") == "ordered" else abi.SCHEME_JOINT
        until = getattr(self, "until", float("inf"))
        return KernelSpec(self.kind, self._k_total, self.mu, self.scale, self.lb, self.ub,
                          self.fixed.astype(np.uint8), scheme=sch, freq=getattr(self, "freq", 1),
                          warmup=getattr(self, "warmup", 0), bw=getattr(self, "bw", 0), until=until,
                          eps=getattr(self, "eps", 1e-4), arate=getattr(self, "arate", 0.234),
                          Sd=getattr(self, "Sd", 0.0) or 0.0, device=device)

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
        names = {1: "kernel_normal", 2: "kernel_normal_reflective", 3: "kernel_adapt", 4: "kernel_ram"}
        return "<fmcmc_kernel %s k=%s>" % (names[self.kind], self.k)


def kernel_normal(mu=0.0, scale=1.0, fixed=False, scheme="joint"):
    return fmcmc_kernel(abi.KERNEL_NORMAL, mu=mu, scale=scale, fixed=fixed, scheme=scheme)


def kernel_normal_reflective(mu=0.0, scale=1.0, lb=-DBL_MAX, ub=DBL_MAX, fixed=False, scheme="joint"):
    return fmcmc_kernel(abi.KERNEL_NORMAL_REFLECTIVE, mu=mu, scale=scale, lb=lb, ub=ub, fixed=fixed,
                        scheme=scheme)


def kernel_adapt(mu=0.0, bw=0, lb=-DBL_MAX, ub=DBL_MAX, freq=1, warmup=500, Sigma=None, Sd=None,
                 eps=1e-4, fixed=False, until=float("inf")):
    if bw > 0 and bw > warmup:
        raise ValueError("The `warmup` parameter must be greater than `bw`.")
    if bw != 0 or freq != 1:
        raise NotImplementedError("device kernel_adapt supports bw = 0 and freq = 1 "
                                  "(SURVEY.md section 8(f) rank 3 lists the windowed variant as next).")
    return fmcmc_kernel(abi.KERNEL_ADAPT, mu=mu, bw=int(bw), lb=lb, ub=ub, freq=int(freq), warmup=int(warmup),
                        Sigma=Sigma, Sd=Sd, eps=eps, fixed=fixed, until=until)


kernel_am = kernel_adapt


def kernel_ram(mu=0.0, eta=None, qfun=None, arate=0.234, freq=1, warmup=0, Sigma=None, eps=1e-4,
               lb=-DBL_MAX, ub=DBL_MAX, fixed=False, until=float("inf"), constr=None):
    if eta is not None or qfun is not None or constr is not None:
        raise NotImplementedError("user-supplied eta/qfun/constr are R closures; the device kernel implements the "
                                  "defaults eta(i,k) = min(1, k i^(-2/3)) and qfun = rt(k, k) (R/kernel_ram.R:67-68).")
    if freq != 1:
        raise NotImplementedError("device kernel_ram supports freq = 1")
    return fmcmc_kernel(abi.KERNEL_RAM, mu=mu, arate=arate, freq=int(freq), warmup=int(warmup), Sigma=Sigma,
                        eps=eps, lb=lb, ub=ub, fixed=fixed, until=until)
