"""Device-level front end of the HIP engine: torch tensors are only used as HBM buffers whose
data_ptr() goes through the C-ABI (include/fmcmc_amd.h, fmcmc_mcmc_run_dev).  One call ==
one MCMC_without_conv_checker over all local chains (R/mcmc.R:485-838)."""
import ctypes as C

import numpy as np
import torch

from . import _abi as abi

DBL_MAX = float(np.finfo(np.float64).max)


def _dev(device=None):
    if not torch.cuda.is_available():
        raise RuntimeError("fmcmc_amd needs an AMD GPU (HIP device); there is no CPU fallback.")
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device(device)


def _t(a, dtype, device):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(device).contiguous()


class DeviceModel:
    """Shared read-only data of a log-posterior family, resident in HBM (fmcmc_model)."""

    def __init__(self, family, X, y, intercept=True, guard=True, prior_div=0.0, device=None):
        self.device = _dev(device)
        if torch.is_tensor(y):
            y = y.to(self.device, torch.float64).contiguous()
        else:
            y = _t(np.asarray(y, dtype=np.float64), torch.float64, self.device)
        n = y.shape[0]
        if X is None:
            Xc, p = None, 0
        else:
            if torch.is_tensor(X):
                X = X.to(self.device, torch.float64)
                if X.ndim == 1:
                    X = X[:, None]
                Xc = X.t().contiguous()  # [p][n] == column-major n x p
            else:
                X = np.asarray(X, dtype=np.float64)
                if X.ndim == 1:
                    X = X[:, None]
                Xc = _t(X.T, torch.float64, self.device)
            p = Xc.shape[0]
            if Xc.shape[1] != n:
                raise ValueError("X has %d rows but y has %d" % (Xc.shape[1], n))
            if p == 0:
                Xc = None
        self.family, self.Xc, self.y, self.n, self.p = family, Xc, y, int(n), int(p)
        self.intercept, self.guard, self.prior_div = int(bool(intercept)), int(bool(guard)), float(prior_div)

    @property
    def k(self):
        if self.family == abi.FAM_GAUSSIAN_LINREG:
            return self.intercept + self.p + 1
        if self.family == abi.FAM_LOGISTIC:
            return self.intercept + self.p
        return 2

    def c(self):
        return abi.Model(self.family, self.p, self.n, self.Xc.data_ptr() if self.Xc is not None else None,
                         self.y.data_ptr(), self.intercept, self.guard, self.prior_div)


class KernelSpec:
    """Expanded (recycled) kernel parameters on the device (fmcmc_kernel)."""

    def __init__(self, kind, k, mu, scale, lb, ub, fixed, scheme=abi.SCHEME_JOINT, freq=1, warmup=0,
                 bw=0, until=float("inf"), eps=1e-4, arate=0.234, Sd=0.0, scheme_seq=None, constr=None, nadapt=4,
                 ram_qfun=0, ram_df=0.0, ram_eta_exp=0.0, device=None):
        self.device = _dev(device)
        self.kind, self.k = int(kind), int(k)
        self.h_fixed = np.ascontiguousarray(np.asarray(fixed, dtype=np.uint8))
        self.kf = int((self.h_fixed == 0).sum())
        # host mirrors of what fmcmc_mcmc_run_dev checks and sizes its launch with (fmcmc_kernel.h_*): with them the
        # entry point only enqueues work
        self.h_lb = np.ascontiguousarray(np.asarray(lb, dtype=np.float64))
        self.h_ub = np.ascontiguousarray(np.asarray(ub, dtype=np.float64))
        self.h_scale = np.ascontiguousarray(np.asarray(scale, dtype=np.float64))
        self.h_seq = None if scheme_seq is None else np.ascontiguousarray(np.asarray(scheme_seq, dtype=np.int32))
        self.mu = _t(mu, torch.float64, self.device)
        self.scale = _t(scale, torch.float64, self.device)
        self.lb = _t(lb, torch.float64, self.device)
        self.ub = _t(ub, torch.float64, self.device)
        self.fixed = _t(self.h_fixed, torch.uint8, self.device)
        self.scheme, self.freq, self.warmup, self.bw = int(scheme), int(freq), int(warmup), int(bw)
        self.until, self.eps, self.arate, self.Sd = float(until), float(eps), float(arate), float(Sd)
        self.nadapt = int(nadapt)
        # kernel_ram's built-in qfun / eta families (fmcmc_kernel.ram_*; zeros = the defaults of R/kernel_ram.R:67-68)
        self.ram_qfun, self.ram_df, self.ram_eta_exp = int(ram_qfun), float(ram_df), float(ram_eta_exp)
        # explicit update sequence: 0-based parameter indices (R/kernel.R:69-92); ram: constr[which., which.] mask
        self.scheme_seq = None if scheme_seq is None else _t(np.asarray(scheme_seq, dtype=np.int32), torch.int32, self.device)
        self.constr = None if constr is None else _t(np.asarray(constr, dtype=np.float64).reshape(self.kf, self.kf),
                                                      torch.float64, self.device)

    @property
    def kz(self):
        """proposal variates per step: one for the single-parameter schemes (R/kernel_normal.R:63)."""
        return 1 if (self.kind in abi.SIMPLE_KERNELS and self.scheme != abi.SCHEME_JOINT) else self.kf

    def c(self):
        return abi.Kernel(self.kind, self.k, self.mu.data_ptr(), self.scale.data_ptr(), self.lb.data_ptr(),
                          self.ub.data_ptr(), self.fixed.data_ptr(), self.scheme, self.freq, self.warmup,
                          self.bw, self.until, self.eps, self.arate, self.Sd,
                          self.scheme_seq.data_ptr() if self.scheme_seq is not None else None,
                          int(self.scheme_seq.numel()) if self.scheme_seq is not None else 0, self.nadapt,
                          self.constr.data_ptr() if self.constr is not None else None,
                          self.h_fixed.ctypes.data, self.h_lb.ctypes.data, self.h_ub.ctypes.data, self.h_scale.ctypes.data,
                          self.h_seq.ctypes.data if self.h_seq is not None else None,
                          self.ram_qfun, 0, self.ram_df, self.ram_eta_exp)


class ChainState:
    """fmcmc_state in HBM: last row of every chain + the kernels' persistent environments."""

    def __init__(self, initial, kf, device=None):
        self.device = _dev(device)
        th = torch.as_tensor(np.ascontiguousarray(initial), dtype=torch.float64) if not torch.is_tensor(initial) else initial
        self.theta0 = th.to(self.device, torch.float64).contiguous().clone()
        Cn = self.theta0.shape[0]
        z = dict(device=self.device)
        self.f0 = torch.zeros(Cn, dtype=torch.float64, **z)
        self.abs_iter = torch.zeros(Cn, dtype=torch.int64, **z)
        self.Sigma = torch.zeros(Cn, kf, kf, dtype=torch.float64, **z)
        self.mean_prev = torch.zeros(Cn, kf, dtype=torch.float64, **z)
        self.have_mean = torch.zeros(Cn, dtype=torch.int32, **z)
        self.nerrors = torch.zeros(Cn, dtype=torch.int32, **z)
        self.scheme_cols = None   # [C][nsteps] int32: plan of scheme = "random" (fmcmc_state.scheme_cols)
        k = self.theta0.shape[1]
        self.mirror_mu = torch.zeros(Cn, k, dtype=torch.float64, **z)      # mirror kernels: adapted mean / scale
        self.mirror_scale = torch.zeros(Cn, k, dtype=torch.float64, **z)
        self.obs_arate = torch.full((Cn, k), float("nan"), dtype=torch.float64, **z)   # (R's obs_arate: a k-vector after warm-up updates)
        self.fresh = 1
        self.step_base = 0

    def c(self, nsteps=None):
        cols = self.scheme_cols
        if cols is not None and nsteps is not None and cols.shape[1] != nsteps:
            if cols.shape[1] < nsteps:   # R: update_sequence[env$i, ] beyond the rows of the kernel's first call
                raise IndexError("subscript out of bounds: the update plan of this kernel has %d rows" % cols.shape[1])
            cols = cols[:, :nsteps].contiguous()
        self._cols_keep = cols
        return abi.State(self.theta0.data_ptr(), self.f0.data_ptr(), self.abs_iter.data_ptr(),
                         self.Sigma.data_ptr(), self.mean_prev.data_ptr(), self.have_mean.data_ptr(),
                         self.nerrors.data_ptr(), self.fresh, 0, cols.data_ptr() if cols is not None else None,
                         self.mirror_mu.data_ptr(), self.mirror_scale.data_ptr(), self.obs_arate.data_ptr())


class SweepResult:
    pass


def kept_rows(nsteps, burnin, thin):
    return (nsteps - burnin) // thin


def sweep(model, kernel, state, nsteps, burnin=0, thin=1, seed=0, chain_base=0,
          want_logpost=True, want_draws=True, want_bits=True, fed_logu=None, fed_z=None,
          stream=None, check=True, into=None, row0=0):
    """Enqueue one sweep (all local chains, nsteps iterations) and return the output tensors.

    into = (samples [C][k][cap], logpost [C][cap] or None, draws [C][k][cap] or None), row0: write the kept rows of this
    call at rows row0.. of a preallocated history (fmcmc_out.ld_rows = cap) instead of allocating; the result then holds
    views of those rows.

    Raises ValueError for argument errors (messages mirror the reference's stop() texts) and
    RuntimeError for chain errors ("fun(par) is undefined", R/mcmc.R:758-765)."""
    L = abi.lib()
    dev = state.device
    Cn, k = state.theta0.shape
    S = kept_rows(nsteps, burnin, thin)
    if S < 0:
        S = 0
    nwords = (nsteps + 31) // 32
    out = SweepResult()
    f64 = dict(dtype=torch.float64, device=dev)
    ld = 0
    if into is not None:
        hs, hl, hd = into
        ld = int(hs.shape[2])
        if row0 + S > ld or hs.shape[0] != Cn or hs.shape[1] != k or not hs.is_contiguous():
            raise ValueError("sweep(into=...): the history holds %s rows, this call writes rows %d..%d" % (ld, row0, row0 + S))
        want_logpost, want_draws = hl is not None, hd is not None
        out.samples = hs[:, :, row0:row0 + S]
        out.logpost = hl[:, row0:row0 + S] if want_logpost else None
        out.draws = hd[:, :, row0:row0 + S] if want_draws else None
    else:
        out.samples = torch.full((Cn, k, max(S, 0)), float("nan"), **f64)
        out.logpost = torch.empty((Cn, S), **f64) if want_logpost else None
        out.draws = torch.empty((Cn, k, S), **f64) if want_draws else None
    out.accept_count = torch.zeros(Cn, dtype=torch.int64, device=dev)
    out.accept_bits = torch.zeros((Cn, nwords), dtype=torch.int32, device=dev) if want_bits else None
    out.status = torch.zeros(Cn, dtype=torch.int32, device=dev)
    out.status_step = torch.zeros(Cn, dtype=torch.int64, device=dev)
    out.status_theta = torch.zeros((Cn, k), **f64)
    rng_mode = abi.RNG_FED if fed_logu is not None else abi.RNG_PHILOX
    crun = abi.Run(Cn, nsteps, burnin, thin, seed & 0xFFFFFFFFFFFFFFFF, chain_base, state.step_base,
                   rng_mode, 0, fed_logu.data_ptr() if fed_logu is not None else None,
                   fed_z.data_ptr() if fed_z is not None else None)
    # (a view's data_ptr() is the address of its first element: row row0 of chain 0, parameter 0)
    cout = abi.Out(out.samples.data_ptr(), out.logpost.data_ptr() if want_logpost else None,
                   out.draws.data_ptr() if want_draws else None, out.accept_count.data_ptr(),
                   out.accept_bits.data_ptr() if want_bits else None, out.status.data_ptr(),
                   out.status_step.data_ptr(), out.status_theta.data_ptr(), ld)
    if kernel.kind in abi.SIMPLE_KERNELS and kernel.scheme == abi.SCHEME_RANDOM and state.scheme_cols is None:
        if rng_mode == abi.RNG_FED:
            raise ValueError("rng_mode = FED with scheme = 'random' needs state.scheme_cols (the plan R drew)")
        state.scheme_cols = torch.zeros((Cn, nsteps), dtype=torch.int32, device=dev)
    cm, ck, cs = model.c(), kernel.c(), state.c(nsteps)
    if stream is None:
        stream = torch.cuda.current_stream(dev)
    with torch.cuda.device(dev):
        rc = L.fmcmc_mcmc_run_dev(C.byref(cm), C.byref(ck), C.byref(crun), C.byref(cs), C.byref(cout),
                                  C.c_void_p(stream.cuda_stream))
    if rc in (abi.ERR_ARG, abi.ERR_UNSUPPORTED):
        raise ValueError(abi.last_error())
    if rc != abi.OK:
        raise RuntimeError("fmcmc_mcmc_run_dev failed (%d): %s" % (rc, abi.last_error()))
    state.fresh = 0
    state.step_base += nsteps
    out.iters = burnin + thin * np.arange(1, S + 1)
    out.thin, out.nsteps, out.burnin = thin, nsteps, burnin
    if check:
        raise_on_chain_error(out, chain_base)
    return out


def rng_stream(state, kernel, nsteps, seed=0, chain_base=0, logu=None, z=None, stream=None):
    """Canonical Philox stream of the next sweep of `state` in HBM (fmcmc_rng_stream_dev); pass the returned
    tensors as fed_logu / fed_z to sweep(): bit-identical to the in-library stream, buffers reusable."""
    L = abi.lib()
    dev = state.device
    Cn = state.theta0.shape[0]
    kz = kernel.kz
    if logu is None:
        logu = torch.empty((Cn, nsteps), dtype=torch.float64, device=dev)
    if z is None:
        z = torch.empty((Cn, nsteps, kz), dtype=torch.float64, device=dev)
    if stream is None:
        stream = torch.cuda.current_stream(dev)
    # the variate family the kernel's own in-library stream draws (mh_engine.hip, launch_sweep): kernel_ram's qfun families
    # (fmcmc_kernel.ram_qfun: rt(k, k) / rnorm(k) / rt(k, df)), U(0,1) for the uniform kernels, N(0,1) otherwise
    if kernel.kind == abi.KERNEL_RAM:
        df = {abi.RAM_QFUN_NORMAL: 0.0, abi.RAM_QFUN_T_DF: kernel.ram_df}.get(kernel.ram_qfun, float(kernel.kf))
    elif kernel.kind in (abi.KERNEL_UNIF, abi.KERNEL_UNIF_REFLECTIVE, abi.KERNEL_UMIRROR):
        df = -1.0
    else:
        df = 0.0
    with torch.cuda.device(dev):
        rc = L.fmcmc_rng_stream_dev(seed & 0xFFFFFFFFFFFFFFFF, state.step_base, chain_base, Cn, nsteps, kz, float(df),
                                    logu.data_ptr(), z.data_ptr(), C.c_void_p(stream.cuda_stream))
    if rc != abi.OK:
        raise RuntimeError("fmcmc_rng_stream_dev failed (%d): %s" % (rc, abi.last_error()))
    return logu, z


def raise_on_chain_error(out, chain_base=0):
    st = out.status.cpu().numpy()
    bad = np.nonzero(st)[0]
    if bad.size:
        if (st == abi.CHAIN_SYNC_TIMEOUT).any():
            raise RuntimeError("fmcmc_amd: a grid-wide hand-over of the observation-sharded evaluation timed out (a device "
                               "fault, or the workgroups of the sweep were not co-resident); the results of this call are "
                               "invalid. FMCMC_AMD_DEBUG=shard=0 selects the chain-sharded kernel.")
        c = int(bad[0])
        step = int(out.status_step[c].item())
        theta = out.status_theta[c].cpu().numpy()
        what = {abi.CHAIN_NAN_LOGPOST: "fun(par) is undefined (NaN).",
                abi.CHAIN_NAN_RATIO: "fun(par) is undefined (f1 - f0 is NaN).",
                abi.CHAIN_NOT_PD: "'Sigma' is not positive definite.",
                abi.CHAIN_BAD_WINDOW: "subscript out of bounds: the rows kernel_adapt(bw / freq) adapts on reach before the "
                                      "first row of this call."}.get(int(st[c]), "chain error.")
        # (R/mcmc.R:759-765 attaches the fun / lb / ub hint to a NaN log-posterior only)
        hint = " Check either -fun- or the -lb- and -ub- parameters." if int(st[c]) in (abi.CHAIN_NAN_LOGPOST, abi.CHAIN_NAN_RATIO) else ""
        raise RuntimeError(
            "%s%s This error ocurred during step i = %d "
            "(chain %d) and proposal parameters theta1 = %s" % (what, hint, step, chain_base + c, np.array2string(theta, precision=4)))
