"""MCMC() and its helpers with the reference's signatures (R/mcmc.R:325-340).

  MCMC / MCMC.default / .mcmc / .mcmc.list   R/mcmc.R:325-479  -> MCMC
  MCMC_without_conv_checker                  R/mcmc.R:485-838  -> MCMC_without_conv_checker
  MCMC_with_conv_checker                     R/mcmc.R:841-1019 -> MCMC_with_conv_checker
  check_initial                              R/checks.R:22-58  -> check_initial
  append_chains                              R/append_chains.R -> append_chains
  MCMC_OUTPUT logpost / draws                R/mcmc.R:822-823, R/mcmc_info.R:351-365 -> get_logpost / get_draws

The per-chain loop itself (R/mcmc.R:720-838) is NOT here: it is the HIP sweep kernel reached through
engine.sweep() -> C-ABI fmcmc_mcmc_run_dev.  `multicore`/`cl` select nothing on a GPU (all chains of
the call already run concurrently); with torch.distributed initialised the chains are sharded over
the ranks in contiguous blocks (one process per GPU) and RCCL is used only by the Gelman check.
"""
import time
import warnings

import numpy as np

from . import _abi as abi
from . import engine
from .kernels import fmcmc_kernel, kernel_normal
from .models import LogPosterior


# ------------------------------------------------------------------------------ coda-like containers
class Mcmc:
    """coda::mcmc: data [niter x nvar] + mcpar = (start, end, thin)."""

    def __init__(self, data, start=1, end=None, thin=1, varnames=None):
        self.data = np.asarray(data, dtype=np.float64)
        if self.data.ndim == 1:
            self.data = self.data[:, None]
        n = self.data.shape[0]
        if end is None:
            end = start + (n - 1) * thin
        if n and (end - start) // thin + 1 != n:
            raise ValueError("Start, end and thin incompatible with data")
        self.mcpar = (int(start), int(end), int(thin))
        self.varnames = list(varnames) if varnames is not None else ["par%d" % (j + 1) for j in range(self.data.shape[1])]
        self.iters = start + thin * np.arange(n)

    niter = property(lambda self: self.data.shape[0])
    nvar = property(lambda self: self.data.shape[1])
    nchain = property(lambda self: 1)
    thin = property(lambda self: self.mcpar[2])
    start = property(lambda self: self.mcpar[0])
    end = property(lambda self: self.mcpar[1])

    def __array__(self, dtype=None):
        return self.data if dtype is None else self.data.astype(dtype)

    def __len__(self):
        return self.niter

    def tail(self, n):
        """utils::tail as fmcmc uses it (R/mcmc.R:362,406): the last n+1 rows."""
        return self.data[-(n + 1):]

    def __repr__(self):
        return "<Mcmc niter=%d nvar=%d mcpar=%s>" % (self.niter, self.nvar, self.mcpar)


class McmcList(list):
    """coda::mcmc.list."""
    nchain = property(lambda self: len(self))
    nvar = property(lambda self: self[0].nvar)
    niter = property(lambda self: self[0].niter)
    thin = property(lambda self: self[0].thin)
    iters = property(lambda self: self[0].iters)

    def as_array(self):
        return np.stack([m.data for m in self])  # [C][S][k]


def check_initial(initial, nchains):
    """R/checks.R:22-58. Returns (matrix [nchains x k], names)."""
    names = None
    if isinstance(initial, dict):
        names, initial = list(initial.keys()), list(initial.values())
    if isinstance(initial, McmcList):
        return np.stack([m.data[-1] for m in initial]), initial[0].varnames
    a = np.asarray(initial, dtype=np.float64)
    if a.ndim <= 1:
        a = np.atleast_1d(a)
        if nchains > 1:
            warnings.warn("While using multiple chains, a single initial point has been passed via `initial`: c(%s). "
                          "The values will be recycled. Ideally you would want to start each chain from different "
                          "locations." % ", ".join(repr(float(v)) for v in a))
        if a.size == 0:
            raise ValueError("The `initial` vector is of length zero.")
        a = np.tile(a, (nchains, 1))
    elif a.ndim != 2:
        raise ValueError("When `initial` is not a numeric vector, it should be a matrix. Right now it is an "
                         "object of class `%s`." % type(initial).__name__)
    elif a.shape[0] != nchains:
        raise ValueError("The number of rows of `initial` (%d) must coincide with the number of chains (%d)."
                         % (a.shape[0], nchains))
    if names is None:
        names = ["par%d" % (j + 1) for j in range(a.shape[1])]
    return np.ascontiguousarray(a), names


def append_chains(*chains):
    """R/append_chains.R:64-143 (iteration labels continue across runs)."""
    chains = [c for c in chains if c is not None and len(c) > 0]
    if not chains:
        raise ValueError("No method available to append these chains.")
    if len(chains) == 1:
        return chains[0]
    if isinstance(chains[0], McmcList):
        ns = [len(c) for c in chains]
        if len(set(ns)) != 1:
            raise ValueError("All mcmc.list objects must have the same number of chains. The passed objects have "
                             "%s respectively." % ", ".join(map(str, ns)))
        return McmcList(append_chains(*[c[i] for c in chains]) for i in range(ns[0]))
    thins = [c.thin for c in chains]
    if len(set(thins)) != 1:
        raise ValueError("All `mcmc` objects have to have the same `thin` parameter.Observed: %s respectively."
                         % ", ".join(map(str, thins)))
    nv = [c.nvar for c in chains]
    if len(set(nv)) != 1:
        raise ValueError("All `mcmc` objects have to have the same number of parameters.Observed: %s respectively."
                         % ", ".join(map(str, nv)))
    data = np.concatenate([c.data for c in chains], axis=0)
    start, thin = chains[0].start, thins[0]
    end = start + (data.shape[0] - 1) * thin
    return Mcmc(data, start=start, end=end, thin=thin, varnames=chains[0].varnames)


# ------------------------------------------------------------------------------ MCMC_OUTPUT (R/mcmc_info.R)
class _Output:
    def __init__(self):
        self.clear()

    def clear(self):
        self.logpost, self.draws, self.elapsed, self.nchains = None, None, None, 0
        self.accept_count, self.kernel = None, None
        self.info = {}            # the call's arguments, R/mcmc.R:443-455 (get_nsteps(), get_seed(), ...)


MCMC_OUTPUT = _Output()


def get_logpost():
    """R/mcmc_info.R:351-355: vector (one chain) or list of vectors."""
    lp = MCMC_OUTPUT.logpost
    if lp is None:
        raise RuntimeError("-logpost- not found in MCMC_OUTPUT.")
    return lp[0] if len(lp) == 1 else lp


def get_draws():
    """R/mcmc_info.R:361-365: proposed states."""
    d = MCMC_OUTPUT.draws
    if d is None:
        raise RuntimeError("-draws- not found in MCMC_OUTPUT.")
    return d[0] if len(d) == 1 else d


def get_elapsed():
    return MCMC_OUTPUT.elapsed


def get_(x):
    """R/mcmc_info.R:301-315: an argument of the last MCMC() call (or logpost / draws / elapsed)."""
    if x in ("logpost", "draws", "elapsed"):
        return {"logpost": get_logpost, "draws": get_draws, "elapsed": get_elapsed}[x]()
    if x not in MCMC_OUTPUT.info:
        raise RuntimeError("-%s- not found in MCMC_OUTPUT." % x)
    return MCMC_OUTPUT.info[x]


def _getter(name):
    def g():
        return get_(name)
    g.__name__ = "get_" + name
    g.__doc__ = "R/mcmc_info.R:317-400: `%s` of the last MCMC() call." % name
    return g


get_initial, get_fun, get_nsteps, get_seed, get_nchains, get_burnin, get_thin, get_kernel, get_multicore, \
    get_conv_checker, get_cl, get_progress, get_chain_id = (_getter(n) for n in (
        "initial", "fun", "nsteps", "seed", "nchains", "burnin", "thin", "kernel", "multicore", "conv_checker", "cl",
        "progress", "chain_id"))


# ------------------------------------------------------------------------------ sharding (one process per GPU)
def shard_bounds(nchains, world, rank):
    """Contiguous chain blocks: chain c lives on rank floor(c * world / nchains)'s block."""
    return (rank * nchains) // world, ((rank + 1) * nchains) // world


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


# ------------------------------------------------------------------------------ device-side result of one call
class DeviceChains:
    """Samples of the local chains, resident in HBM: samples [C][k][S] (+ logpost, draws).

    With `capacity` rows allocated up front (MCMC_with_conv_checker: the kept rows of all bulks, R/mcmc.R:926-947) the
    buffers are [C][k][capacity] and `nrows` of them are filled; samples / logpost / draws are views of the filled part
    (their row stride stays `capacity`, which is what fmcmc_gelman_partial_dev takes as S)."""

    def __init__(self, samples, logpost, draws, iters, thin, names, chain_base, nchains_total, nrows=None):
        self._samples, self._logpost, self._draws = samples, logpost, draws
        self.nrows = int(samples.shape[-1]) if nrows is None else int(nrows)
        self.iters, self.thin, self.names = np.asarray(iters), thin, names
        self.chain_base, self.nchains_total = chain_base, nchains_total

    samples = property(lambda self: self._samples[:, :, :self.nrows])
    logpost = property(lambda self: None if self._logpost is None else self._logpost[:, :self.nrows])
    draws = property(lambda self: None if self._draws is None else self._draws[:, :, :self.nrows])
    capacity = property(lambda self: int(self._samples.shape[-1]))

    @classmethod
    def allocate(cls, nchains_local, k, capacity, thin, names, chain_base, nchains_total, device, want_logpost=True,
                 want_draws=True):
        import torch
        f64 = dict(dtype=torch.float64, device=device)
        return cls(torch.full((nchains_local, k, capacity), float("nan"), **f64),
                   torch.empty((nchains_local, capacity), **f64) if want_logpost else None,
                   torch.empty((nchains_local, k, capacity), **f64) if want_draws else None,
                   np.zeros(0, dtype=np.int64), thin, names, chain_base, nchains_total, nrows=0)

    def extend(self, iters_of_call):
        """The rows of one more call were written behind the filled part (engine.sweep(into=...)): labels continue
        (R/append_chains.R:113-142)."""
        it = np.asarray(iters_of_call)
        if it.size:
            self.iters = it if self.iters.size == 0 else np.concatenate(
                [self.iters, it - it[0] + self.iters[-1] + self.thin])
            self.nrows += int(it.size)

    def append(self, other):
        import torch
        iters = np.concatenate([self.iters, other.iters - other.iters[0] + self.iters[-1] + self.thin])
        cat = lambda a, b: None if a is None or b is None else torch.cat([a, b], dim=-1)
        return DeviceChains(cat(self.samples, other.samples), cat(self.logpost, other.logpost),
                            cat(self.draws, other.draws), iters, self.thin, self.names, self.chain_base,
                            self.nchains_total)

    def to_host(self):
        s = self.samples.cpu().numpy()  # [C][k][S]
        start, end = (int(self.iters[0]), int(self.iters[-1])) if self.iters.size else (1, 0)
        chains = [Mcmc(s[c].T.copy(), start=start, end=end, thin=self.thin, varnames=self.names)
                  for c in range(s.shape[0])]
        return chains[0] if (len(chains) == 1 and self.nchains_total == 1) else McmcList(chains)


def _validate_common(nsteps, nchains, burnin, thin, multicore):
    if multicore and nchains == 1:
        raise ValueError("When `multicore = TRUE`, `nchains` should be greater than 1.")
    if nchains < 1:
        raise ValueError("`nchains` must be an integer greater than 1.")
    if burnin >= nsteps:
        raise ValueError("-burnin- (%d) cannot be >= than -nsteps- (%d)." % (burnin, nsteps))
    if thin >= nsteps:
        raise ValueError("-thin- (%d) cannot be > than -nsteps- (%d)." % (thin, nsteps))
    if thin < 1:
        raise ValueError("-thin- should be >= 1.")


def _run_call(initial_local, fun, nsteps, burnin, thin, kernel, seed, chain_base, nchains_total, names,
              device, want_logpost=True, want_draws=True, history=None, fed=None):
    """One MCMC_without_conv_checker over the local chains -> DeviceChains (history: append the rows to it instead)."""
    import torch
    gm = fun.device_model(device)
    kernel._init(initial_local.shape[1])
    if kernel._spec is None or kernel._spec.device != gm.device:
        kernel._spec = kernel.spec(gm.device)
    if initial_local.shape[0] == 0:
        # a rank without chains (nchains < number of ranks): nothing to launch, but it still takes part in the checker's
        # all-reduce with an empty partial
        if history is not None:
            history.extend(burnin + thin * np.arange(1, engine.kept_rows(nsteps, burnin, thin) + 1))
            return history
        k, S = initial_local.shape[1], max(engine.kept_rows(nsteps, burnin, thin), 0)
        f64 = dict(dtype=torch.float64, device=gm.device)
        dc = DeviceChains(torch.empty((0, k, S), **f64), torch.empty((0, S), **f64) if want_logpost else None,
                          torch.empty((0, k, S), **f64) if want_draws else None,
                          burnin + thin * np.arange(1, S + 1), thin, names, chain_base, nchains_total)
        dc.accept_count = torch.zeros(0, dtype=torch.int64, device=gm.device)
        return dc
    st = kernel.state_for(initial_local, gm.device)
    fkw = {}
    if fed is not None:
        # FMCMC_RNG_FED: the caller supplies the variates of this call in the order the reference draws them (per chain:
        # log(runif(nsteps)), then the kernel's draws step by step, chains one after the other: R/mcmc.R:643-673,726)
        logu, z = fed(initial_local.shape[0], nsteps, kernel._spec.kz, kernel)
        fkw = dict(fed_logu=torch.as_tensor(np.ascontiguousarray(logu, dtype=np.float64)).to(gm.device),
                   fed_z=torch.as_tensor(np.ascontiguousarray(z, dtype=np.float64)).to(gm.device))
    if history is not None:
        out = engine.sweep(gm, kernel._spec, st, nsteps, burnin=burnin, thin=thin, seed=seed, chain_base=chain_base,
                           want_bits=False, into=(history._samples, history._logpost, history._draws), row0=history.nrows,
                           **fkw)
        history.extend(out.iters)
        history.accept_count = out.accept_count
        return history
    out = engine.sweep(gm, kernel._spec, st, nsteps, burnin=burnin, thin=thin, seed=seed, chain_base=chain_base,
                       want_logpost=want_logpost, want_draws=want_draws, want_bits=False, **fkw)
    dc = DeviceChains(out.samples, out.logpost, out.draws, out.iters, thin, names, chain_base, nchains_total)
    dc.accept_count = out.accept_count
    return dc


def MCMC_without_conv_checker(initial, fun, nsteps, nchains=1, burnin=0, thin=1, kernel=None, multicore=False,
                              conv_checker=None, cl=None, progress=False, chain_id=1, seed=0, device=None,
                              _return_device=False, keep_logpost=True, keep_draws=True, fed=None):
    """R/mcmc.R:485-838 for all chains at once."""
    if kernel is None:
        kernel = kernel_normal()
    if not isinstance(fun, LogPosterior):
        raise TypeError("-fun- must be one of the engine's closed-form families (gaussian_linreg, logistic, "
                        "iid_normal): an arbitrary closure cannot run inside the fused GPU kernel.")
    if not isinstance(kernel, fmcmc_kernel):
        raise TypeError("-kernel- must be an fmcmc_kernel (kernel_normal, kernel_normal_reflective, kernel_adapt, kernel_ram).")
    init, names = check_initial(initial, nchains)
    _validate_common(nsteps, nchains, burnin, thin, multicore)
    if init.shape[1] != fun.k:
        raise ValueError("Incorrect length of -initial-: the model has %d parameters, got %d." % (fun.k, init.shape[1]))
    dist, rank, world = _dist()
    lo, hi = shard_bounds(nchains, world, rank)
    dc = _run_call(init[lo:hi], fun, nsteps, burnin, thin, kernel, seed, lo, nchains, names, device,
                   want_logpost=keep_logpost, want_draws=keep_draws, fed=fed)
    _store_output(dc, kernel)
    return dc if _return_device else dc.to_host()


def _store_output(dc, kernel):
    lp = dc.logpost.cpu().numpy() if dc.logpost is not None else None
    dr = dc.draws.cpu().numpy() if dc.draws is not None else None
    MCMC_OUTPUT.logpost = [lp[c] for c in range(lp.shape[0])] if lp is not None else None
    MCMC_OUTPUT.draws = [dr[c].T.copy() for c in range(dr.shape[0])] if dr is not None else None
    MCMC_OUTPUT.nchains = dc.samples.shape[0]
    MCMC_OUTPUT.kernel = kernel


def MCMC_with_conv_checker(initial, fun, nsteps, nchains, burnin, thin, kernel, multicore, conv_checker, cl=None,
                           progress=False, chain_id=1, seed=0, device=None, _return_device=False, verbose=True,
                           keep_logpost=True, keep_draws=True, fed=None):
    """R/mcmc.R:841-1019: run in bulks of `freq`, restart every chain from its last row, stop when the
    checker says so."""
    if conv_checker is None:
        raise ValueError("The convergence checker for this call cannot be null.")
    freq = getattr(conv_checker, "freq", None)
    if freq is None:
        freq = nsteps // 2
        warnings.warn("The -conv_checker- function has no freq attribute. Default value set to be %d" % freq)
    if freq * 2 > nsteps:
        freq = 0
    if freq > 0:
        bulks = [freq] * ((nsteps - burnin) // freq)
        if (nsteps - burnin) % freq:
            bulks.append((nsteps - burnin) - sum(bulks))
    else:
        bulks = [nsteps]
    bulks[0] += burnin
    if kernel is None:
        kernel = kernel_normal()
    if not isinstance(fun, LogPosterior):
        raise TypeError("-fun- must be one of the engine's closed-form families (gaussian_linreg, logistic, "
                        "iid_normal): an arbitrary closure cannot run inside the fused GPU kernel.")
    if not isinstance(kernel, fmcmc_kernel):
        raise TypeError("-kernel- must be an fmcmc_kernel (kernel_normal, kernel_normal_reflective, kernel_adapt, kernel_ram).")
    init, names = check_initial(initial, nchains)
    dist, rank, world = _dist()
    lo, hi = shard_bounds(nchains, world, rank)
    init_local = init[lo:hi]
    conv_checker.flush()
    converged = False
    free = None
    # the history of all bulks, allocated once: every bulk's kept rows are written behind the previous ones by the sweep
    # itself (fmcmc_out.ld_rows) -- no per-bulk concatenation of a history that reaches GBs at config C4
    capacity = sum(engine.kept_rows(nb, burnin if bi == 0 else 0, thin) for bi, nb in enumerate(bulks))
    gdev = fun.device_model(device).device
    ans = DeviceChains.allocate(hi - lo, init.shape[1], capacity, thin, names, lo, nchains, gdev,
                                want_logpost=keep_logpost, want_draws=keep_draws)
    for bi, nb in enumerate(bulks):
        if bi > 0:
            burnin = 0
            # initial <- ans[niter(ans), ]: the last KEPT row of every chain (R/mcmc.R:908-911), which with thin > 1 is not
            # the last row the loop visited
            # (a first bulk can end without a kept row -- burnin > 0 and freq < thin < freq + burnin pass the argument checks --
            #  where R's `ans[niter(ans), ]` has no row to take; the chains then continue from the state the kernel carries,
            #  their true last row, instead of from the NaN prefill of the history)
            init_local = ans._samples[:, :, ans.nrows - 1] if ans.nrows > 0 else kernel._state.theta0
        _validate_common(nb, nchains, burnin, thin, multicore)
        _run_call(init_local, fun, nb, burnin, thin, kernel, seed, lo, nchains, names, device, history=ans, fed=fed)
        if free is None:
            free = np.nonzero(~kernel.fixed)[0]
        converged = conv_checker.check_device(ans, free)
        msg = conv_checker.msg
        total = sum(bulks[:bi + 1])
        if converged:
            if verbose and rank == 0:
                print("Convergence has been reached with %d steps. %s(%d final count of samples)."
                      % (total, (msg + " ") if msg else "", ans.samples.shape[-1]))
            break
        elif verbose and rank == 0:
            print("No convergence yet (steps count: %d). %sTrying with the next bulk." % (total, (msg + " ") if msg else ""))
    if not converged and verbose and rank == 0:
        print("No convergence reached after %d steps (%d final count of samples)." % (sum(bulks[:bi + 1]), ans.samples.shape[-1]))
    ans.converged = converged
    _store_output(ans, kernel)
    return ans if _return_device else ans.to_host()


_seed_counter = [int(time.time_ns()) & 0xFFFFFFFF]


def MCMC(initial, fun, nsteps, *, seed=None, nchains=1, burnin=0, thin=1, kernel=None, multicore=False,
         conv_checker=None, cl=None, progress=False, chain_id=1, device=None, _return_device=False,
         keep_logpost=True, keep_draws=True, fed=None):
    """Drop-in for fmcmc::MCMC (R/mcmc.R:325-340).

    initial: vector, [nchains x k] matrix, or a previous result (Mcmc: its last `nchains` rows,
    R/mcmc.R:344-378; McmcList: each chain's last row, :382-422).  fun: gaussian_linreg / logistic /
    iid_normal object.  seed: Philox key (None: a fresh one per call; with torch.distributed every rank uses rank 0's).
    keep_logpost / keep_draws = False: do not record MCMC_OUTPUT's logpost / draws (R always does, R/mcmc.R:822-823; at
    config C4 the draws alone are 2 GB per GPU).  fed: callable (nchains, nsteps, kz, kernel) -> (logu [C][nsteps],
    z [C][nsteps][kz]) supplying the variates of every call instead of the Philox stream (fmcmc_run.rng_mode = FED; one
    process only): fed R's own Mersenne-Twister stream the engine retraces fmcmc's printed outputs."""
    MCMC_OUTPUT.clear()
    t0 = time.time()
    if isinstance(initial, Mcmc):
        initial = initial.tail(nchains - 1)
        if initial.shape[0] == 1:
            initial = initial[0]
    elif isinstance(initial, McmcList):
        if nchains != len(initial):
            raise ValueError("The parameter `nchains` must equal the number of chains passed by `initial`.")
    if seed is None:
        _seed_counter[0] = (_seed_counter[0] * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        seed = _seed_counter[0] & 0x7FFFFFFFFFFFFFFF
        dist, rank, world = _dist()
        if dist is not None:
            # the RNG is keyed by (seed, GLOBAL chain id): every rank must use the same key or sharding would change the chains
            import torch
            dev = fun.device_model(device).device if isinstance(fun, LogPosterior) else "cpu"
            t = torch.tensor([seed], dtype=torch.int64, device=dev)
            dist.broadcast(t, src=0)
            seed = int(t.item())
    if kernel is None:
        kernel = kernel_normal()
    MCMC_OUTPUT.info = dict(initial=initial, fun=fun, nsteps=nsteps, seed=seed, nchains=nchains, burnin=burnin, thin=thin,
                            kernel=kernel, multicore=multicore, conv_checker=conv_checker, cl=cl, progress=progress,
                            chain_id=chain_id)
    if conv_checker is not None:
        ans = MCMC_with_conv_checker(initial, fun, nsteps, nchains, burnin, thin, kernel, multicore, conv_checker,
                                     cl, progress, chain_id, seed=seed, device=device, _return_device=_return_device,
                                     keep_logpost=keep_logpost, keep_draws=keep_draws, fed=fed)
    else:
        ans = MCMC_without_conv_checker(initial, fun, nsteps, nchains, burnin, thin, kernel, multicore, None, cl,
                                        progress, chain_id, seed=seed, device=device, _return_device=_return_device,
                                        keep_logpost=keep_logpost, keep_draws=keep_draws, fed=fed)
    MCMC_OUTPUT.elapsed = time.time() - t0
    return ans
