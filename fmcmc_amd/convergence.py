"""convergence_gelman (R/convergence.R:191-246): Gelman-Rubin auto-stop.

The reference hands the whole accumulated mcmc.list to coda::gelman.diag on every check.  Here the
samples stay in HBM: a HIP kernel reduces the window (second half of the history, coda's
autoburnin) of every local chain to its mean and covariance, a second one sums those over the local
chains in a fixed order, ONE all-reduce(sum) of 1 + 5p + 2p^2 doubles joins the GPUs (RCCL over
xGMI; the only collective of the engine), and fmcmc_gelman_finish forms W, B, psrf and mpsrf.
"""
import ctypes as C

import numpy as np

from . import _abi as abi


def _window(iters):
    """coda's autoburnin: if start(x) < end(x)/2 keep iterations >= end/2 + 1."""
    iters = np.asarray(iters)
    if iters.size == 0:          # a bulk that kept no row (burnin / thin): nothing to test yet
        return 0, 0
    start, end = iters[0], iters[-1]
    if start < end / 2:
        row0 = int(np.searchsorted(iters, end / 2 + 1, side="left"))
    else:
        row0 = 0
    return row0, int(iters.size - row0)


class convergence_gelman:
    def __init__(self, freq=1000, threshold=1.10, check_invariant=True):
        self.freq, self.threshold, self.check_invariant = int(freq), float(threshold), bool(check_invariant)
        self.flush()

    def flush(self):
        """convergence_data_flush (R/convergence.R:119-165): LAST_CONV_CHECK store."""
        self.history = []  # (end iteration, value, psrf)
        self.msg = ""
        self.last = None

    # -------- device path used by MCMC_with_conv_checker
    def check_device(self, chains, cols, group=None):
        """chains: DeviceChains of the LOCAL chains. Returns the same logical on every rank."""
        import torch
        import torch.distributed as dist
        L = abi.lib()
        samples = chains.samples             # view of the filled rows; the buffer's row stride is its capacity
        Cn, k, S = samples.shape
        stride = chains.capacity
        dev = samples.device
        p = int(len(cols))
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        total_chains = chains.nchains_total if distributed else Cn
        self.msg = ""
        if total_chains < 2:
            raise ValueError("Convergence test with the Gelman is only available when `nchains` > 1L.")
        row0, N = _window(chains.iters)
        if N < 2:
            return False
        cols_d = torch.as_tensor(np.asarray(cols, dtype=np.int32)).to(dev)
        plen = int(L.fmcmc_gelman_partial_len(p))
        partial = torch.zeros(plen + 2, dtype=torch.float64, device=dev)
        # centre = first kept row of global chain 0 (broadcast through the same all-reduce pattern)
        center = torch.zeros(p, dtype=torch.float64, device=dev)
        if chains.chain_base == 0 and Cn > 0:
            center.copy_(samples[0, cols_d.long(), row0])
        if distributed:
            dist.all_reduce(center, op=dist.ReduceOp.SUM, group=group)
        if Cn > 0 and p > abi.MAX_K_WAVE:
            # more free parameters than the MFMA window reduction tiles (64): the same per-chain statistics -- window mean relative
            # to the centre, window covariance -- and their sums over the chains with torch on the device (64 < k <= 128 is the
            # big-k kernel's territory: one chain per workgroup, speed is not the point there)
            # in blocks of chains, so that the gathered window, its centred copy and the products stay bounded (1024 chains x k = 100 x
            # N = 5000 rows were three 4 GB tensors at once): per-chain statistics are independent, the sums over the chains add up
            per_chain = 3 * p * N * 8
            blk_c = max(1, min(Cn, (256 << 20) // max(per_chain, 1)))
            xbm = torch.empty((Cn, p), dtype=torch.float64, device=dev)
            s2 = torch.empty((Cn, p), dtype=torch.float64, device=dev)
            sum_xx = torch.zeros((p, p), dtype=torch.float64, device=dev)
            sum_S = torch.zeros((p, p), dtype=torch.float64, device=dev)
            for c0 in range(0, Cn, blk_c):
                X = samples[c0:c0 + blk_c][:, cols_d.long(), row0:row0 + N] - center[None, :, None]
                xb_b = X.mean(dim=2)
                X -= xb_b[:, :, None]
                Sc = X @ X.transpose(1, 2) / float(N - 1)
                xbm[c0:c0 + blk_c] = xb_b
                s2[c0:c0 + blk_c] = Sc.diagonal(dim1=1, dim2=2)
                sum_xx += (xb_b[:, :, None] * xb_b[:, None, :]).sum(0)
                sum_S += Sc.sum(0)
                del X, Sc
            partial[0] = float(Cn)
            o = 1
            for blk in (xbm.sum(0), sum_xx.reshape(-1), sum_S.reshape(-1), s2.sum(0), (s2 * s2).sum(0),
                        (s2 * xbm).sum(0), (s2 * xbm * xbm).sum(0)):
                partial[o:o + blk.numel()] = blk
                o += blk.numel()
            if self.check_invariant:
                xb = xbm + center
                partial[plen] = float(N) * xb.sum()
                partial[plen + 1] = ((N - 1.0) * s2 + float(N) * xb * xb).sum()
        elif Cn > 0:
            work = torch.empty(int(L.fmcmc_gelman_work_len(Cn, p)), dtype=torch.float64, device=dev)
            with torch.cuda.device(dev):
                rc = L.fmcmc_gelman_partial_dev(samples.data_ptr(), Cn, k, stride, row0, N, cols_d.data_ptr(), p,
                                                center.data_ptr(), work.data_ptr(), partial.data_ptr(),
                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            if rc != abi.OK:
                raise RuntimeError("fmcmc_gelman_partial_dev failed (%d)" % rc)
            if self.check_invariant:  # rm_invariant: sd of ALL entries (R/convergence.R:171-173)
                # sum and sum of squares of the whole window from the per-chain means / variances the reduction just left in
                # `work` (an indexed copy of the window itself was 1 GB and ~1 ms per check at config C4)
                wk = work.view(Cn, p + p * p)
                xb = wk[:, :p] + center
                s2 = wk[:, p:].reshape(Cn, p, p).diagonal(dim1=1, dim2=2)
                partial[plen] = float(N) * xb.sum()
                partial[plen + 1] = ((N - 1.0) * s2 + float(N) * xb * xb).sum()
        if distributed:
            dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)  # the engine's only collective
        ph = partial.cpu().numpy()
        return self._finish(ph, p, N, int(chains.iters[-1]), plen, total_chains)

    def _finish(self, ph, p, N, end_iter, plen, total_chains):
        L = abi.lib()
        if self.check_invariant:
            cnt = total_chains * p * N
            var_all = (ph[plen + 1] - ph[plen] ** 2 / cnt) / max(cnt - 1, 1)
            if var_all < 1e-10:
                return False
        psrf = np.empty(p)
        mps = C.c_double()
        part = np.ascontiguousarray(ph[:plen])
        rc = L.fmcmc_gelman_finish(part.ctypes.data_as(C.POINTER(C.c_double)), p, N,
                                   psrf.ctypes.data_as(C.POINTER(C.c_double)), C.byref(mps))
        if rc != abi.OK:
            import warnings
            warnings.warn("At %d `gelman.diag` failed to be computed. Will skip and try with the next batch." % end_iter)
            return False
        val = mps.value if p > 1 else float(psrf[0])
        self.history.append((end_iter, val, psrf.copy()))
        self.last = val
        self.msg = "Gelman-Rubin's R: %.4f." % val
        return bool(val < self.threshold)

    # -------- R-style call on a host mcmc.list
    def __call__(self, x):
        import torch
        from .mcmc import McmcList, DeviceChains
        if not isinstance(x, McmcList) or len(x) < 2:
            raise ValueError("Convergence test with the Gelman is only available when `nchains` > 1L.")
        arr = np.ascontiguousarray(x.as_array().transpose(0, 2, 1))  # [C][k][S]
        dc = DeviceChains(torch.as_tensor(arr).cuda(), None, None, x.iters, x.thin, x[0].varnames, 0, len(x))
        return self.check_device(dc, np.arange(arr.shape[1]))


# ================================================================================================
# Single-chain checkers (R/convergence.R:248-360).  Their arithmetic lives in coda (not under the reference tree);
# restated from coda's published algorithms: spectrum0.ar = AR(p) fit by Yule-Walker / Levinson-Durbin with AIC order
# selection (stats::ar.yw), geweke.diag, heidel.diag (Cramer-von Mises statistic of the Brownian bridge, pcramer series).
# They are host-side (one chain, a few thousand numbers): the chain is copied out of HBM once per check.
# ================================================================================================
def _ar_yw(y):
    """stats::ar(y, aic = TRUE) (ar.yw.default): returns (coefficients, var.pred, order)."""
    y = np.asarray(y, dtype=np.float64)
    n = y.size
    order_max = int(min(n - 1, np.floor(10 * np.log10(n))))
    x = y - y.mean()
    r = np.array([np.dot(x[:n - l], x[l:]) / n for l in range(order_max + 1)])   # acf(type = "covariance"): divisor n
    if r[0] == 0:
        raise FloatingPointError("zero-variance series")
    coefs = np.zeros((order_max + 1, order_max + 1))
    vars_ = np.empty(order_max + 1)
    vars_[0] = r[0]
    for m in range(1, order_max + 1):            # Levinson-Durbin (eureka)
        acc = r[m] - np.dot(coefs[m - 1, 1:m], r[m - 1:0:-1])
        phi = acc / vars_[m - 1]
        coefs[m, m] = phi
        coefs[m, 1:m] = coefs[m - 1, 1:m] - phi * coefs[m - 1, m - 1:0:-1]
        vars_[m] = vars_[m - 1] * (1 - phi * phi)
    aic = n * np.log(vars_) + 2 * np.arange(order_max + 1) + 2
    order = int(np.argmin(aic))
    var_pred = vars_[order] * n / (n - (order + 1))
    return coefs[order, 1:order + 1].copy(), float(var_pred), order


def spectrum0_ar(y):
    """coda::spectrum0.ar for one series: (spectral density at frequency zero, AR order)."""
    y = np.asarray(y, dtype=np.float64)
    n = y.size
    z = np.arange(1, n + 1, dtype=np.float64)
    if n < 3:
        raise FloatingPointError("series too short")
    A = np.stack([np.ones(n), z], axis=1)
    res = y - A @ np.linalg.lstsq(A, y, rcond=None)[0]
    sd = np.sqrt(np.sum((res - res.mean()) ** 2) / (n - 1))
    if sd < 1.5e-8:                               # identical(all.equal(sd(residuals), 0), TRUE)
        return 0.0, 0
    ar, var_pred, order = _ar_yw(y)
    return var_pred / (1 - ar.sum()) ** 2, order


def _window_rows(iters, start=None, end=None):
    """stats::window on an mcmc object: rows whose iteration label lies in [start, end]."""
    iters = np.asarray(iters, dtype=np.float64)
    lo = 0 if start is None else int(np.searchsorted(iters, start - 1e-9, side="left"))
    hi = iters.size if end is None else int(np.searchsorted(iters, end + 1e-9, side="right"))
    return lo, hi


def geweke_diag(data, iters, frac1=0.1, frac2=0.5):
    """coda::geweke.diag: z-scores of mean(first frac1) - mean(last frac2), variances from spectrum0.ar."""
    data = np.asarray(data, dtype=np.float64)
    start, end = float(iters[0]), float(iters[-1])
    xstart = (start, np.floor(end - frac2 * (end - start)))
    xend = (np.ceil(start + frac1 * (end - start)), end)
    means, variances = [], []
    for s, e in zip(xstart, xend):
        lo, hi = _window_rows(iters, s, e)
        w = data[lo:hi]
        variances.append(np.array([spectrum0_ar(w[:, j])[0] for j in range(w.shape[1])]) / w.shape[0])
        means.append(w.mean(axis=0))
    with np.errstate(divide="ignore", invalid="ignore"):
        return (means[0] - means[1]) / np.sqrt(variances[0] + variances[1])


def _pcramer(q, eps=1e-5):
    """Distribution function of the Cramer-von Mises statistic (coda::heidel.diag's pcramer)."""
    from scipy.special import gamma, kv
    log_eps = np.log(eps)
    total = 0.0
    for k in range(4):
        zc = gamma(k + 0.5) * np.sqrt(4 * k + 1) / (gamma(k + 1) * np.pi ** 1.5 * np.sqrt(q))
        u = (4 * k + 1) ** 2 / (16 * q)
        total += 0.0 if u > -log_eps else zc * np.exp(-u) * kv(0.25, u)
    return float(total)


def heidel_diag(data, iters, eps=0.1, pvalue=0.05):
    """coda::heidel.diag: rows (stest, start, pvalue, htest, mean, halfwidth) per variable."""
    data = np.asarray(data, dtype=np.float64)
    iters = np.asarray(iters, dtype=np.float64)
    n_all = data.shape[0]
    out = np.full((data.shape[1], 6), np.nan)
    start, end = iters[0], iters[-1]
    start_vec = np.arange(start, end / 2 + 1e-9, n_all / 10.0) if n_all / 10.0 > 0 else np.array([start])
    for j in range(data.shape[1]):
        Y, it = data[:, j], iters
        lo, _ = _window_rows(it, end / 2)
        S0 = spectrum0_ar(Y[lo:])[0]
        converged, I = False, np.nan
        for st in start_vec:
            lo, _ = _window_rows(it, st)
            Y, it = Y[lo:], it[lo:]
            n = Y.size
            ybar = Y.mean()
            B = np.cumsum(Y) - ybar * np.arange(1, n + 1)
            with np.errstate(divide="ignore", invalid="ignore"):
                I = float(np.sum(B * B / (n * S0)) / n)
            converged = bool(np.isfinite(I) and _pcramer(I) < 1 - pvalue)
            if converged:
                break
        S0ci = spectrum0_ar(Y)[0]
        halfwidth = 1.96 * np.sqrt(S0ci / n)
        passed = bool(np.isfinite(halfwidth) and abs(halfwidth / ybar) <= eps)
        if (not converged) or (not np.isfinite(I)) or (not np.isfinite(halfwidth)):
            out[j] = [float(converged), np.nan, 1 - _pcramer(I) if np.isfinite(I) else np.nan, np.nan, np.nan, np.nan]
        else:
            out[j] = [1.0, it[0], 1 - _pcramer(I), float(passed), ybar, halfwidth]
    return out


class _SingleChainChecker:
    """Common plumbing of the single-chain checkers: LAST_CONV_CHECK-style history, rm_invariant, device entry."""
    name = ""

    def __init__(self, freq=1000, check_invariant=True):
        self.freq, self.check_invariant = int(freq), bool(check_invariant)
        self.flush()

    def flush(self):
        self.history, self.msg, self.last = [], "", None

    def check_device(self, chains, cols, group=None):
        if chains.nchains_total > 1:
            raise ValueError(self._multi_msg)
        data = chains.samples[0].cpu().numpy().T[:, np.asarray(cols)]      # [S][p]
        return self._check(data, chains.iters)

    def __call__(self, x):
        from .mcmc import Mcmc, McmcList
        if isinstance(x, McmcList):
            if len(x) > 1:
                raise ValueError(self._multi_msg)
            x = x[0]
        if not isinstance(x, Mcmc):
            raise TypeError("expected an Mcmc object")
        return self._check(x.data, x.iters)

    def _prepare(self, data):
        data = np.asarray(data, dtype=np.float64)
        if self.check_invariant and data.size > 1 and np.std(data, ddof=1) < 1e-10:    # rm_invariant, R/convergence.R:169-186
            return None
        return data


class convergence_geweke(_SingleChainChecker):
    """R/convergence.R:248-292.  As in the reference, the quantity compared with `threshold` is d = 1 - 2 pnorm(-|z|)
    (one minus the two-sided p-value): the check returns TRUE iff every d > threshold."""
    _multi_msg = "The `geweke` convergence check is only available with runs of a single chain."

    def __init__(self, freq=1000, threshold=0.025, check_invariant=True, frac1=0.1, frac2=0.5):
        super().__init__(freq, check_invariant)
        self.threshold, self.frac1, self.frac2 = float(threshold), frac1, frac2

    def _check(self, data, iters):
        import warnings
        from math import erfc
        data = self._prepare(data)
        try:
            if data is None:
                raise FloatingPointError("invariant chain")
            z = geweke_diag(data, iters, self.frac1, self.frac2)
        except (FloatingPointError, np.linalg.LinAlgError, ValueError, ZeroDivisionError):
            warnings.warn("At %d `geweke.diag` failed to be computed. Will skip and try with the next batch." % len(iters))
            return False
        self.history.append((int(iters[-1]), z.copy()))
        fin = z[np.isfinite(z)]
        self.last = float(fin.mean()) if fin.size else float("nan")
        self.msg = "avg Geweke's Z: %.4f." % self.last
        d = np.array([1 - erfc(abs(v) / np.sqrt(2.0)) if np.isfinite(v) else np.nan for v in z])   # 1 - 2 pnorm(-|z|)
        if np.any(~np.isfinite(d)):
            return False
        return bool(np.all(d > self.threshold))


class convergence_heildel(_SingleChainChecker):
    """R/convergence.R:295-344: converged iff the stationarity and the half-width tests pass for every parameter."""
    _multi_msg = "The -heidel- convergence check is only available with runs of a single chain."

    def __init__(self, freq=1000, check_invariant=True, eps=0.1, pvalue=0.05):
        super().__init__(freq, check_invariant)
        self.eps, self.pvalue = eps, pvalue

    def _check(self, data, iters):
        import warnings
        data = self._prepare(data)
        try:
            if data is None:
                raise FloatingPointError("invariant chain")
            d = heidel_diag(data, iters, self.eps, self.pvalue)
        except (FloatingPointError, np.linalg.LinAlgError, ValueError, ZeroDivisionError):
            warnings.warn("At %d -coda::heidel.diag- failed to be computed. Will skip and try with the next batch." % len(iters))
            return False
        self.history.append((int(iters[-1]), d.copy()))
        self.last = float(np.nanmean(d[:, 2]))
        self.msg = "Heidel's Avg. pval: %.2f" % self.last
        tests = d[:, [0, 3]]
        if np.any(~np.isfinite(tests)):
            return False
        return bool(np.all(tests == 1))


class convergence_auto:
    """R/convergence.R:346-368: Gelman-Rubin when there are several chains, Geweke otherwise."""

    def __init__(self, freq=1000):
        self.freq = int(freq)
        self._gelman, self._geweke = convergence_gelman(freq), convergence_geweke(freq)
        self._used = self._geweke

    def flush(self):
        self._gelman.flush()
        self._geweke.flush()

    def check_device(self, chains, cols, group=None):
        self._used = self._gelman if chains.nchains_total > 1 else self._geweke
        return self._used.check_device(chains, cols, group)

    def __call__(self, x):
        from .mcmc import McmcList
        self._used = self._gelman if isinstance(x, McmcList) and len(x) > 1 else self._geweke
        return self._used(x)

    history = property(lambda self: self._used.history)
    msg = property(lambda self: self._used.msg)
    last = property(lambda self: self._used.last)
