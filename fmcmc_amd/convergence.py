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
        samples = chains.samples
        Cn, k, S = samples.shape
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
        if Cn > 0:
            work = torch.empty(int(L.fmcmc_gelman_work_len(Cn, p)), dtype=torch.float64, device=dev)
            with torch.cuda.device(dev):
                rc = L.fmcmc_gelman_partial_dev(samples.data_ptr(), Cn, k, S, row0, N, cols_d.data_ptr(), p,
                                                center.data_ptr(), work.data_ptr(), partial.data_ptr(),
                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            if rc != abi.OK:
                raise RuntimeError("fmcmc_gelman_partial_dev failed (%d)" % rc)
            if self.check_invariant:  # rm_invariant: sd of ALL entries (R/convergence.R:171-173)
                win = samples[:, cols_d.long(), row0:]
                partial[plen] = win.sum()
                partial[plen + 1] = (win * win).sum()
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
