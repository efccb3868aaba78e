#!/usr/bin/env python3
"""Benchmark of the hot path: MH samples/s (chains x iterations / s).

Default = the headline, BASELINE.json configs[1]: 1024 chains x 5-parameter Gaussian linear regression, n = 10,000,
kernel_normal, 1 MI355X.  `--config c3|c4|c5` runs the other GPU configs of BASELINE.md section 4 with their PER-GPU share
of chains (C4 512, C5 1024) and prints the same JSON shape.

One "step" = one full sweep of the hot path over the workload: all chains of the rank advance the config's number of MH
iterations, writing ans / logpost / draws like the reference (R/mcmc.R:728-734,822-823).  C2, C3, C5: ONE fused kernel
launch per step.  C4: the config runs under convergence_gelman(freq = 1000), so a step is ten 1000-iteration bulks appended
to one preallocated history, each followed by the Gelman check (device reduction + the engine's only collective, an
all-reduce over RCCL when N > 1); the threshold is set so that no check stops the run and every step does the same work.
Inputs (X, y, initial states) are resident in HBM before the timed region.

  python bench.py [--gpus N --steps K --warmup W] [--config c2|c3|c4|c5] [--scaling weak|strong]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU.  Under a launcher (WORLD_SIZE set) the rank count must equal --gpus; WITHOUT one,
`python bench.py --gpus N` starts its N rank processes itself (launch_ranks: fresh children with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* set, before this process touches a GPU; the analogue of makePSOCKcluster, R/mcmc.R:536-545) and relays
rank 0's line.  `config.ranks_seen` lists every rank's device ordinal and architecture as gathered through the process group.
--scaling weak (default): every rank runs the config's per-GPU share, chain ids continue across ranks, no data-path
collective.  --scaling strong (C2 / C3): the config's 1024 chains are divided over the ranks.

The line reports what was measured in THIS run: roofline.kernel is fmcmc_last_kernel() (and the run fails if the dispatcher
did not pick the kernel the config is tuned for), roofline.kernel_ms comes from HIP events on the launch stream.  roofline.traffic cannot be measured from inside the process
(PMC counters): it is the figure of the newest COMMITTED PMC pass of this same command, labelled as such in
roofline.traffic_source (file, the commit that pass measured, its kernel and duration) and shown only while the dispatcher
still picks the kernel that pass measured; null otherwise (profiles/README.md says how those passes are taken).
The default (headline) invocation on one GPU additionally runs a short full-size sweep of C3, C4 and C5 and reports them
under `configs` (value, kernel, kernel_ms, bound, frac each), so that one driver-timed run covers all four GPU configs.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_TFLOPS = 78.6   # MI355X fp64 vector = fp64 matrix peak (256 CU x 128 flop/clk x 2.4 GHz)
PEAK_HBM_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
CHAIN_SEED = 1215


# ------------------------------------------------------------------------------------------------ the four GPU configs
class Config:
    """BASELINE.md section 4 / SURVEY.md 8(d): data seeds 20260101 + config number, chain RNG seed 1215."""

    def __init__(self, name):
        self.name = name
        if name in ("c2", "c3"):
            self.num, self.n, self.p, self.k = (2, 10000, 3, 5) if name == "c2" else (3, 10000, 3, 5)
            self.chains, self.iters, self.thin = 1024, 10000, 1
            self.family = "linreg"
            self.kernel_name = "kernel_normal(scale=0.02)" if name == "c2" else "kernel_adapt() (warmup 500)"
            self.expect_kernel = "mfma" if name == "c2" else "spec"
            self.flops = self.n * (2 * self.p + 3)          # E n (2(p-1)+3), p-1 = 3 covariates
            self.flops_note = "SURVEY 8(d): n (2 x 3 + 3) = 9.0e4 per sample"
            self.bulk = None
        elif name == "c4":
            self.num, self.n, self.p, self.k = 4, 10000, 48, 50
            self.chains, self.iters, self.thin = 512, 10000, 1
            self.family = "linreg"
            self.kernel_name = "kernel_ram() + convergence_gelman(freq=1000)"
            self.expect_kernel = "wide-dataflow"
            # unbounded kernel_ram evaluates the log-posterior ONCE per step (the un-reflected proposal IS the proposal):
            # E = 1 -> n (2 x 48 + 3) + 3 k^2 (S u, factor update).  SURVEY 8(d) prices E = 2 (1.98e6); both are reported.
            self.flops = self.n * (2 * self.p + 3) + 3 * self.k * self.k
            self.flops_note = "executed: E = 1 evaluation per step (unbounded kernel_ram) = n (2 x 48 + 3) + 3 k^2 = 9.98e5; SURVEY 8(d) counts E = 2: 1.98e6"
            self.bulk = 1000
        elif name == "c5":
            self.num, self.n, self.p, self.k = 5, 100000, 5, 6
            self.chains, self.iters, self.thin = 1024, 5000, 10
            self.family = "logistic"
            self.kernel_name = "kernel_normal_reflective(scale=0.01, lb=-5, ub=5), thin 10"
            self.expect_kernel = "logistic-shadow"
            self.flops = self.n * (2 * self.p + 8)          # exp and log1p counted as 1 flop each
            self.flops_note = "SURVEY 8(d): n (2 x 5 + 8) = 1.8e6 per sample, transcendentals counted as one flop each"
            self.bulk = None
        else:
            raise SystemExit("unknown --config %s" % name)
        self.data_seed = 20260100 + self.num

    def workload(self, chains, chain_base):
        rng = np.random.default_rng(self.data_seed)
        n, p, k = self.n, self.p, self.k
        X = rng.standard_normal((n, p))
        if self.name in ("c2", "c3"):
            beta = np.array([3.0, 2.0, -1.0, 0.5])
            y = beta[0] + X @ beta[1:] + 4.0 * rng.standard_normal(n)
            base = np.array([0.0, 0.0, 0.0, 0.0, float(np.std(y, ddof=1))])
            jit = 0.1
        elif self.name == "c4":
            beta = rng.standard_normal(p + 1)
            y = beta[0] + X @ beta[1:] + 2.0 * rng.standard_normal(n)
            base = np.concatenate([beta, [2.0]])
            jit = 0.01
        else:
            beta = np.array([-1.0, 0.5, -0.5, 0.25, -0.25, 1.0])
            y = (rng.uniform(size=n) < 1.0 / (1.0 + np.exp(-(beta[0] + X @ beta[1:])))).astype(np.float64)
            base = beta.copy()
            jit = 0.01
        irng = np.random.default_rng(self.data_seed + 1)
        init = base[None, :] + jit * irng.standard_normal((chain_base + chains, k))[chain_base:]
        if self.family == "linreg":
            init[:, -1] = np.abs(init[:, -1])
        return X, y, np.ascontiguousarray(init)

    # kernel parameters shared by the engine's KernelSpec and the oracle's Kernel
    def kernel_args(self):
        k = self.k
        if self.name == "c2":
            return dict(kind=1, scale=0.02)
        if self.name == "c3":
            return dict(kind=3, warmup=500)
        if self.name == "c4":
            return dict(kind=4)
        return dict(kind=2, scale=0.01, lb=-5.0, ub=5.0)


def device_objects(cfg, E, abi, X, y, dev):
    big = E.DBL_MAX
    k, ka = cfg.k, cfg.kernel_args()
    fam = abi.FAM_GAUSSIAN_LINREG if cfg.family == "linreg" else abi.FAM_LOGISTIC
    gm = E.DeviceModel(fam, X, y, device=dev) if cfg.family == "linreg" else \
        E.DeviceModel(fam, X, y, intercept=True, guard=False, prior_div=8.0, device=dev)
    gk = E.KernelSpec(ka["kind"], k, np.zeros(k), np.full(k, ka.get("scale", 1.0)), np.full(k, ka.get("lb", -big)),
                      np.full(k, ka.get("ub", big)), np.zeros(k, np.uint8), warmup=ka.get("warmup", 0), device=dev)
    return gm, gk


def oracle_objects(cfg, O, X, y):
    ka = cfg.kernel_args()
    if cfg.family == "linreg":
        m = O.Model(O.FAM_LINREG, X, y)
    else:
        m = O.Model(O.FAM_LOGISTIC, X, y, intercept=True, guard=False, prior_div=8.0)
    kw = {a: ka[a] for a in ("scale", "lb", "ub", "warmup") if a in ka}
    return m, O.Kernel({1: O.K_NORMAL, 2: O.K_NORMAL_REFLECTIVE, 3: O.K_ADAPT, 4: O.K_RAM}[ka["kind"]], cfg.k, **kw)


def host_cores():
    """Threads this process may really use: the affinity mask, capped by the cgroup CPU quota (os.cpu_count() ignores both)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(cfg, seconds_budget=24.0):
    """The oracle (PHILOX / canonical mode = the GPU's exact outputs) timed on the host cores on a bounded sample of the SAME
    workload: (a) all usable host threads, one block of chains per thread -- the analogue of fmcmc's multicore = TRUE with
    min(nchains, detectCores()) workers (R/mcmc.R:539-541); (b) one thread.  Both samples are sized
    from the STEADY-STATE cost per iteration (the difference of two short runs: thread start-up, the first evaluation and
    the page faults of the outputs cancel), so that each really runs for its share of the budget."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    cores = host_cores()
    X, y, init0 = cfg.workload(cores, 0)
    m, k = oracle_objects(cfg, O, X, y)

    def run(init, lo, iters):
        O.run(m, k, init, nsteps=iters, thin=cfg.thin if iters > cfg.thin else 1, seed=CHAIN_SEED, chain_base=lo, want_draws=True)

    def timed(fn):
        t = time.perf_counter(); fn(); return time.perf_counter() - t

    def steady(fn, lo_it, hi_it):           # seconds per iteration of one chain, fixed costs removed
        a, b = timed(lambda: fn(lo_it)), timed(lambda: fn(hi_it))
        return max(b - a, 1e-9) / (hi_it - lo_it)

    def all_threads(inits, per_thread, iters):
        with ThreadPoolExecutor(cores) as ex:             # ctypes releases the GIL: one oracle call per host thread
            list(ex.map(lambda tix: run(inits[tix * per_thread:(tix + 1) * per_thread], tix * per_thread, iters), range(cores)))

    def refine(per, fn):                    # a second two-point difference, ~1.5 s long, sized from the first estimate
        it = int(max(80, min(cfg.iters, 1.5 / per)))
        return steady(fn, it // 4, it)

    # (b) one thread: ~1/4 of the budget (whole chains when a chain fits)
    per1 = steady(lambda it: run(init0[:1], 0, it), 11, 61)
    per1 = refine(per1, lambda it: run(init0[:1], 0, it))
    want1 = 0.25 * seconds_budget
    it1 = int(max(50, min(cfg.iters, want1 / per1)))
    ch1 = int(max(1, min(cores, want1 / (per1 * it1))))
    dt1 = timed(lambda: run(init0[:ch1], 0, it1))
    single = {"value": ch1 * (it1 - 1) / dt1, "unit": "MH samples/s", "cores": 1,
              "sample": "%d chain(s) x %d iterations" % (ch1, it1), "seconds": dt1}
    # (a) all threads busy: steady-state cost under full load (SMT and memory contention included), then size the sample
    per = steady(lambda it: all_threads(init0, 1, it), 11, 61)
    per = refine(per, lambda it: all_threads(init0, 1, it))
    want = 0.75 * seconds_budget
    if per * cfg.iters <= want:                       # whole chains: several per thread
        per_thread, iters = int(max(1, min(256, want / (per * cfg.iters)))), cfg.iters
    else:                                             # a chain is longer than the budget: a prefix of its iterations
        per_thread, iters = 1, int(max(50, want / per))
    chains = cores * per_thread
    _, _, init = cfg.workload(chains, 0)
    dt = timed(lambda: all_threads(init, per_thread, iters))
    value = chains * (iters - 1) / dt
    return {"value": value, "unit": "MH samples/s", "cores": cores, "kind": "port", "seconds": dt,
            "cores_note": "threads used = len(os.sched_getaffinity(0)) capped by the cgroup cpu.max quota (os.cpu_count() = %d)" % (os.cpu_count() or 0),
            "parallel_efficiency": value / (cores * single["value"]),
            "sample": "%d chains x %d iterations of the same workload (%d chains per host thread), oracle/fmcmc_oracle.c in "
                      "PHILOX/canonical mode: a C restatement of fmcmc's R loop that produces the GPU's exact outputs; "
                      "optimistic vs interpreted R" % (chains, iters, per_thread),
            "single_thread": single}


KERNEL_FN = {"mfma": "mh_sweep_mfma", "wide-dataflow": "mh_sweep_wide2", "spec": "mh_sweep_spec", "streamed-logistic": "mh_sweep_kernel", "logistic-shadow": "mh_sweep_logit2", "logistic-sharded": "mh_sweep_kernel"}
# what the dominant kernel of a config is bound by: C2 / C4 evaluate on the matrix cores, C3 / C5 on the fp64 VALU.  Either
# way the peak is the ONE fp64 datapath of MI355X (fp64 matrix peak == fp64 vector peak, 78.6 TFLOP/s).
BOUND = {"c2": "mfma", "c3": "valu", "c4": "mfma", "c5": "valu"}
DEFAULT_STEPS = {"c2": 240, "c3": 140, "c4": 20, "c5": 6}
EXTRA_STEPS = {"c3": 20, "c4": 3, "c5": 2}     # the short sweeps of the other configs inside the default (headline) invocation


def run_config(cfg, chains, iters, steps, warmup, world, rank, dev, dist, torch, E, abi):
    """W warm-up steps, then exactly K timed steps of one config between barriers; returns what was measured."""
    thin, k = cfg.thin, cfg.k
    chain_base = rank * chains
    X, y, init = cfg.workload(chains, chain_base)
    gm, gk = device_objects(cfg, E, abi, X, y, dev)
    init_d = torch.as_tensor(init).to(dev)
    S = iters // thin
    picked = []
    cur = {"s": None}
    ev_mid = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    ev_chk = []
    bulks = None
    chk = None

    if cfg.name == "c4":
        # the config's own loop: bulks of 1000 iterations appended to one preallocated history (fmcmc_out.ld_rows), the Gelman
        # check behind each (fmcmc_gelman_partial_dev + all-reduce + fmcmc_gelman_finish); draws are not recorded (2 GB)
        from fmcmc_amd.mcmc import DeviceChains
        from fmcmc_amd.convergence import convergence_gelman
        chk = convergence_gelman(cfg.bulk, threshold=0.0)      # R-hat >= 1 > 0: no check stops the run
        bulks = [cfg.bulk] * (iters // cfg.bulk) + ([iters % cfg.bulk] if iters % cfg.bulk else [])
        hist = DeviceChains.allocate(chains, k, S, thin, None, chain_base, world * chains, dev, want_logpost=True, want_draws=False)
        free = np.arange(k)

        def one_step():
            st = E.ChainState(init_d, k, device=dev)
            hist.nrows, hist.iters = 0, np.zeros(0, dtype=np.int64)
            chk.flush()
            out = None
            for nb in bulks:
                out = E.sweep(gm, gk, st, nb, thin=thin, seed=CHAIN_SEED, chain_base=chain_base, want_bits=False, check=False,
                              into=(hist._samples, hist._logpost, None), row0=hist.nrows)
                picked.append(abi.last_kernel())
                hist.extend(out.iters)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                chk.check_device(hist, free)
                e1.record()
                ev_chk.append((e0, e1))
            return out
    else:
        # the canonical Philox stream of a sweep is generated by its own small kernel into reusable HBM buffers where the
        # sweep kernel consumes a materialised stream (C2, C3: bit-identical to letting the library do both)
        streamed_rng = cfg.name in ("c2", "c3")
        if streamed_rng:
            logu_buf = torch.empty((chains, iters), dtype=torch.float64, device=dev)
            z_buf = torch.empty((chains, iters, gk.kz), dtype=torch.float64, device=dev)

        def one_step():
            st = E.ChainState(init_d, k, device=dev)
            kw = {}
            if streamed_rng:
                E.rng_stream(st, gk, iters, seed=CHAIN_SEED, chain_base=chain_base, logu=logu_buf, z=z_buf)
                kw = dict(fed_logu=logu_buf, fed_z=z_buf)
            if cur["s"] is not None:
                ev_mid[cur["s"]].record()
            out = E.sweep(gm, gk, st, iters, thin=thin, seed=CHAIN_SEED, chain_base=chain_base, want_logpost=True,
                          want_draws=True, want_bits=False, check=False, **kw)
            picked.append(abi.last_kernel())
            return out

    out = None
    for _ in range(warmup):
        out = one_step()
    torch.cuda.synchronize()
    ev_chk.clear()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    # collectives issued inside the timed region (the engine's only one is the all-reduce of the Gelman check): counted, so
    # that the line says whether the measured steps contained it
    ncoll = {"all_reduce": 0}
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(*a, **kw):
        ncoll["all_reduce"] += 1
        return real_all_reduce(*a, **kw)

    if world > 1:
        dist.all_reduce = counting_all_reduce
    t0 = time.perf_counter()
    for s in range(steps):
        cur["s"] = s
        ev[s][0].record()            # same stream the kernels are launched on
        out = one_step()
        ev[s][1].record()
    torch.cuda.synchronize()
    dist.all_reduce = real_all_reduce
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    bases = [chain_base]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        b = torch.zeros(world, dtype=torch.float64, device=dev)     # what every rank really used as its first global chain id
        b[rank] = float(chain_base)
        dist.all_reduce(b, op=dist.ReduceOp.SUM)
        bases = [int(v) for v in b.tolist()]
    assert int(out.status.abs().sum().item()) == 0, "a chain reported an error"
    # the library's DEFAULT path of the same sweep (no materialised stream handed in: the library fills its own per step
    # window, mh_engine.hip launch_sweep), timed beside the fed form the steps above used: what a plain MCMC() call gets
    default_ms = None
    if cfg.name in ("c2", "c3") and not getattr(cfg, "no_default_path", False):
        nd = max(2, min(steps, 12))
        d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        outs_d = None
        for i in range(nd + 1):
            if i == 1:
                d0.record()
            st = E.ChainState(init_d, k, device=dev)
            outs_d = E.sweep(gm, gk, st, iters, thin=thin, seed=CHAIN_SEED, chain_base=chain_base, want_logpost=True,
                             want_draws=True, want_bits=False, check=False)
            picked.append(abi.last_kernel())
        d1.record()
        torch.cuda.synchronize()
        default_ms = d0.elapsed_time(d1) / nd
        assert torch.equal(outs_d.samples, out.samples), "default path and fed path differ"
    bad = sorted(set(n for n in picked if n != cfg.expect_kernel))
    if bad and chains == cfg.chains and iters == cfg.iters and not getattr(cfg, "any_kernel", False):
        raise SystemExit("bench.py --config %s: the dispatcher picked %s, the config is measured on '%s'" % (cfg.name, bad, cfg.expect_kernel))
    # kernel duration from HIP events on the launch stream
    step_ms = float(np.mean([ev[s][0].elapsed_time(ev[s][1]) for s in range(steps)]))
    if cfg.name == "c4":
        chk_ms = float(np.sum([a.elapsed_time(b) for a, b in ev_chk])) / steps      # all checks of a step
        kern_ms, extra = step_ms - chk_ms, {"gelman_checks_ms_per_step": chk_ms, "bulks_per_step": len(bulks),
                                            "launches_per_step": len(bulks), "rhat_last": chk.last}
    elif cfg.name in ("c2", "c3"):
        kern_ms = float(np.mean([ev_mid[s].elapsed_time(ev[s][1]) for s in range(steps)]))   # sweep kernel (+ output memsets)
        extra = {"rng_fill_kernel_ms": step_ms - kern_ms, "default_path_ms_per_step": default_ms,
                 "default_path_note": "the same sweep through the library's own step windows (stream filled per window), bit-equal outputs"}
    else:
        kern_ms, extra = step_ms, {}
    samples_per_step = chains * (iters - 1)
    acc = float(out.accept_count.double().mean().item()) / ((bulks[-1] if cfg.name == "c4" else iters) - 1)
    return {"cfg": cfg, "chains": chains, "iters": iters, "steps": steps, "warmup": warmup, "world": world, "S": S,
            "elapsed": elapsed, "value": world * samples_per_step * steps / elapsed, "ms_per_step": 1e3 * elapsed / steps,
            "kern_ms": kern_ms, "extra": extra, "picked": sorted(set(picked)), "accept_rate": acc,
            "samples_per_step": samples_per_step, "chain_base": chain_base, "chain_bases": bases,
            "all_reduce_calls_in_timed_steps": ncoll["all_reduce"]}


def roofline_block(res, traffic_from=None):
    cfg, kern_ms, chains, iters = res["cfg"], res["kern_ms"], res["chains"], res["iters"]
    k, S = cfg.k, res["S"]
    ach_tflops = res["samples_per_step"] * cfg.flops / (kern_ms * 1e-3) / 1e12
    out_bytes = chains * S * ((1 if cfg.name == "c4" else 2) * k + 1) * 8
    # HBM bytes per launch of the sweep kernel: PMC counters cannot be read from inside this process, so `traffic` is NOT a
    # measurement of this run: it is the figure of the newest committed PMC pass of this same command
    # (profiles/latest_pmc_<config>.json, written by tools/profile_bench.sh, which records the commit it measured), shown only
    # while the dispatcher still picks the kernel that pass measured, and labelled as such in traffic_source; null otherwise.
    traffic, tsrc = None, None
    here = os.path.dirname(os.path.abspath(__file__))
    tfile = traffic_from or os.path.join(here, "profiles", "latest_pmc_%s.json" % cfg.name)
    if os.path.exists(tfile):
        trec = json.load(open(tfile))
        if traffic_from or all(KERNEL_FN.get(pk, pk) in trec.get("kernel", "") for pk in res["picked"]):
            traffic = trec.get("hbm_bytes_per_launch")
            tsrc = {"measured_in_this_run": False, "file": os.path.relpath(tfile, here), "pass_of_commit": trec.get("head"),
                    "pass_kernel": trec.get("kernel"), "pass_kernel_avg_ns": trec.get("kernel_avg_ns_rocprof")}
    rl = {"bound": BOUND[cfg.name],
          "pipe": "fp64 datapath (MFMA and VALU share it: MI355X fp64 matrix peak == fp64 vector peak); this config evaluates on the %s"
                  % ("matrix cores (v_mfma_f64)" if BOUND[cfg.name] == "mfma" else "fp64 VALU"),
          "achieved": ach_tflops, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
          "frac": ach_tflops / PEAK_FP64_TFLOPS, "traffic": traffic, "traffic_source": tsrc,
          "kernel": res["picked"][0] if len(res["picked"]) == 1 else res["picked"],
          "kernel_ms": kern_ms, "flops_per_sample": cfg.flops, "flops_note": cfg.flops_note,
          "hbm": {"achieved_GBps": out_bytes / (kern_ms * 1e-3) / 1e9, "peak_GBps": PEAK_HBM_GBS,
                  "algorithmic_bytes_per_step": out_bytes}}
    # the whole step the caller pays for (C2 / C3: + rng_fill_kernel; C4: + the Gelman checks), beside the sweep kernel's own fraction
    rl["frac_step"] = res["samples_per_step"] * cfg.flops / (res["ms_per_step"] * 1e-3) / 1e12 / PEAK_FP64_TFLOPS
    launches = len(BULKS_OF(cfg, iters)) if cfg.name == "c4" else 1
    rl["algorithmic_bytes_per_launch"] = out_bytes / launches
    rl["traffic_ratio"] = (traffic / (out_bytes / launches)) if traffic else None
    rl.update(res["extra"])
    return rl


def BULKS_OF(cfg, iters):
    return [cfg.bulk] * (iters // cfg.bulk) + ([iters % cfg.bulk] if iters % cfg.bulk else [])


def workload_text(res):
    cfg = res["cfg"]
    return "configs[%d] (%s): %d chains/GPU x %d-param %s n=%d, %s, nsteps=%d, outputs ans+logpost%s" % (
        cfg.num - 1, cfg.name.upper(), res["chains"], cfg.k, "Gaussian linreg" if cfg.family == "linreg" else "logistic regression",
        cfg.n, cfg.kernel_name, res["iters"], "" if cfg.name == "c4" else "+draws")


def launch_ranks(args_list, n, backend):
    """`python bench.py --gpus N` without a launcher around it: start N FRESH rank processes (the analogue of
    makePSOCKcluster(ncores), R/mcmc.R:536-545) BEFORE this process imports torch or touches a GPU, one per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; relay rank 0's JSON line and the worst return code.  Nothing is
    re-executed in place: the parent only waits."""
    import socket
    import subprocess
    with socket.socket() as s:      # (a free port of this moment; the ranks' rendezvous has a timeout, see main)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FMCMC_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + args_list, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # every rank is watched: when one dies, the others (blocked in the rendezvous or in a collective until the backend's
    # timeout) are ended at once, and nothing outlives this process
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read() or ""), daemon=True)
    reader.start()
    rcs = [None] * n
    t_fail = None
    try:
        while any(rc is None for rc in rcs):
            for r, pr in enumerate(procs):
                if rcs[r] is None:
                    rcs[r] = pr.poll()
            if any(rc not in (None, 0) for rc in rcs):
                t_fail = t_fail or time.time()
                if time.time() - t_fail > 3.0:      # (grace: ranks that fail for the same reason report it themselves)
                    break
            time.sleep(0.05)
    finally:
        for r, pr in enumerate(procs):
            if pr.poll() is None:
                pr.terminate()
        for r, pr in enumerate(procs):
            try:
                rc = pr.wait(timeout=20)
            except subprocess.TimeoutExpired:
                pr.kill()
                rc = pr.wait()
            if rcs[r] is None:
                rcs[r] = rc if rc else (0 if all(x in (None, 0) for x in rcs) else -15)
    reader.join(timeout=5)
    sys.stdout.write("".join(out0))
    sys.stdout.flush()
    worst = max(rcs, key=abs) if any(rcs) else 0
    if worst:
        sys.stderr.write("bench.py: rank return codes %s\n" % rcs)
    return worst if 0 <= worst < 256 else 1


def install_segv_trace():
    """FMCMC_SEGV_TRACE=1 (diagnosis): native backtrace on SIGSEGV (tools/segv_trace.c), e.g. for the crash of a rocprofv3-profiled
    run inside an exit handler; FMCMC_SEGV_TRACE_FILE names the file it goes to (default stderr)."""
    if os.environ.get("FMCMC_SEGV_TRACE") != "1":
        return
    import ctypes
    so = os.path.join(ROOT, "tools", "exp_bin", "libsegv_trace.so")
    if os.path.exists(so):
        lib = ctypes.CDLL(so)
        lib.fmcmc_segv_trace_install(os.environ.get("FMCMC_SEGV_TRACE_FILE", "").encode())


def main():
    install_segv_trace()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: about 5 s of GPU time for the config: c2 240, c3 140, c4 20, c5 6)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--iters", type=int, default=0, help="MH iterations per sweep (default: the config's nsteps)")
    ap.add_argument("--chains", type=int, default=0, help="chains per GPU (default: the config's per-GPU share)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="headline invocation only: skip the short C3 / C4 / C5 sweeps reported under `configs`")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend for N > 1 (nccl = RCCL; gloo moves the tensors through the host: tests on one GPU)")
    ap.add_argument("--no-default-path", action="store_true", help="skip the extra leg that times the library's default path (profiling passes: keeps the sweep kernel's launches uniform)")
    ap.add_argument("--any-kernel", action="store_true", help="diagnosis: do not insist on the kernel the config is tuned for (FMCMC_AMD_DEBUG knobs)")
    ap.add_argument("--traffic-from", default=None, help="JSON of a PMC pass of this same command (hbm_bytes_per_launch)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank runs the config's per-GPU share (C2/C3 1024 chains per rank); strong: C2/C3's 1024 chains "
                         "are divided over the ranks (north_star: '1024 chains ... at 1/2/4/8 MI355X'); C4/C5 keep their per-GPU share")
    args = ap.parse_args()
    cfg = Config(args.config)
    cfg.any_kernel = args.any_kernel
    cfg.no_default_path = args.no_default_path
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:                       # no launcher: be one (before anything here has initialised a GPU)
            sys.exit(launch_ranks(sys.argv[1:], args.gpus, args.backend))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ["WORLD_SIZE"]))
    if args.steps <= 0:   # a timed region of about five seconds (one step = one sweep of the config's nsteps iterations)
        args.steps = DEFAULT_STEPS[cfg.name]

    import torch
    import torch.distributed as dist
    from fmcmc_amd import engine as E, _abi as abi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    if args.backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())      # several ranks may share the one GPU of a test box
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get("FMCMC_BENCH_RDZV_TIMEOUT", "180")))   # (a rank that never arrives must not hang the others for the backend's default half hour)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group("gloo", timeout=tmo)

    chains = args.chains or cfg.chains
    scaling = "weak"
    strong_ok = world > 1 and not args.chains and cfg.chains % world == 0
    if args.scaling == "strong" and world > 1 and not args.chains:
        if cfg.chains % world:
            raise SystemExit("bench.py --scaling strong: %d chains do not divide over %d ranks" % (cfg.chains, world))
        chains, scaling = cfg.chains // world, "strong"
    iters = args.iters or cfg.iters
    # who is really there: every rank's device ordinal and architecture, gathered through the process group
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "device": int(torch.cuda.current_device()), "arch": str(getattr(props, "gcnArchName", "?")).split(":")[0],
          "pid": os.getpid()}
    ranks_seen = [me]
    if world > 1:
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, me)
    res = run_config(cfg, chains, iters, args.steps, args.warmup, world, rank, dev, dist, torch, E, abi)
    # N > 1 GPUs: BOTH scalings in one invocation (every config since round 5).  The reference scales a FIXED number of chains over its workers
    # (R/mcmc.R:536-641), i.e. strong scaling -- 1024 / N chains per GPU, the latency form of the kernels below 1024 --, the
    # contract's line is per-GPU work fixed (weak).  `value` / `scaling` are the leg --scaling names; `scalings` holds both.
    scalings = {scaling: {"value": res["value"], "chains_per_gpu": chains, "total_chains": chains * world, "ms_per_step": res["ms_per_step"],
                          "kernel": res["picked"]}}
    if strong_ok:
        other = "strong" if scaling == "weak" else "weak"
        oc = cfg.chains // world if other == "strong" else cfg.chains
        ores = run_config(cfg, oc, iters, max(3, args.steps // 4), 1, world, rank, dev, dist, torch, E, abi)
        scalings[other] = {"value": ores["value"], "chains_per_gpu": oc, "total_chains": oc * world, "ms_per_step": ores["ms_per_step"],
                           "kernel": ores["picked"], "steps": ores["steps"]}
    # one GPU: what a GPU of a strong-scaled call of this config would see -- 1024 / G chains, measured HERE -- and the curve
    # that follows from it (G x the samples/s of 1024 / G chains on one GPU: chains never interact, R/mcmc.R:590-673)
    projection = None
    if world == 1 and not args.chains and not args.iters and not args.no_extra_configs:
        projection = {"note": "measured on this one GPU with %d / G chains; value_G = G x samples/s of that run (%s)" %
                              (cfg.chains, "the all-reduce of a Gelman check, 1 + 5p + 2p^2 doubles per 1000 iterations, not included" if cfg.name == "c4"
                               else "no data-path collective exists"),
                      "1": {"chains_per_gpu": cfg.chains, "ms_per_step": res["ms_per_step"], "value": res["value"], "kernel": res["picked"]}}
        for G in (2, 4, 8):
            pr = run_config(cfg, cfg.chains // G, iters, max(3, args.steps // 8), 1, 1, 0, dev, dist, torch, E, abi)
            projection[str(G)] = {"chains_per_gpu": cfg.chains // G, "ms_per_step": pr["ms_per_step"], "value": G * pr["value"],
                                  "kernel": pr["picked"], "speedup_vs_1": G * pr["value"] / res["value"]}

    # the other GPU configs of BASELINE.md section 4 as short full-size sweeps, so that the driver's clock covers them too
    # (headline invocation on one GPU only; `--config cX` gives each its own full line, `--gpus N` its scaling)
    extras = {}
    if cfg.name == "c2" and world == 1 and not args.no_extra_configs and not args.chains and not args.iters:
        for name in ("c3", "c4", "c5"):
            xc = Config(name)
            t_x = time.perf_counter()
            xr = run_config(xc, xc.chains, xc.iters, EXTRA_STEPS[name], 1, 1, 0, dev, dist, torch, E, abi)
            rl = roofline_block(xr)
            extras[name] = {"workload": workload_text(xr), "value": xr["value"], "unit": "MH samples/s", "steps": xr["steps"], "warmup": 1,
                            "ms_per_step": xr["ms_per_step"], "accept_rate": xr["accept_rate"], "kernel": rl["kernel"],
                            "kernel_ms": rl["kernel_ms"], "bound": rl["bound"], "achieved": rl["achieved"], "peak": rl["peak"],
                            "frac": rl["frac"], "frac_step": rl["frac_step"], "flops_per_sample": rl["flops_per_sample"], "traffic": rl["traffic"],
                            "traffic_ratio": rl["traffic_ratio"], "algorithmic_bytes_per_launch": rl["algorithmic_bytes_per_launch"],
                            "traffic_source": rl["traffic_source"], "wall_s_incl_setup": None}
            for key in ("gelman_checks_ms_per_step", "rng_fill_kernel_ms", "default_path_ms_per_step"):
                if key in rl:
                    extras[name][key] = rl[key]
            extras[name]["wall_s_incl_setup"] = time.perf_counter() - t_x

    if rank == 0:
        metric = "MH samples/sec (chains x iters / s), 1024 chains, 5-param linreg n=10k" if cfg.name == "c2" else \
                 "MH samples/sec (chains x iters / s), config %s" % cfg.name.upper()
        if world > 1 and scaling == "weak":   # (per-GPU work fixed: the job holds N x the config's chains)
            metric += " [weak scaling: %d x %d chains]" % (world, chains)
        line = {
            "metric": metric,
            "value": res["value"], "unit": "MH samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload_text(res),
                       "chains_per_gpu": chains, "iters_per_step": iters, "thin": cfg.thin, "accept_rate": res["accept_rate"],
                       "parallelism": "chains sharded, %d rank(s), %s" % (world, "one all-reduce of 1 + 5p + 2p^2 doubles per Gelman check"
                                                                          if cfg.name == "c4" else "no data-path collective"),
                       "backend": (args.backend if world > 1 else None), "chain_base_of_rank": res["chain_bases"],
                       "world": world, "ranks_seen": ranks_seen,
                       "launched_by": ("bench.py itself (%d fresh rank processes)" % world) if os.environ.get("FMCMC_BENCH_SPAWNED")
                                      else ("an external launcher" if world > 1 else "single process"),
                       "all_reduce_calls_in_timed_steps": res["all_reduce_calls_in_timed_steps"]},
            "roofline": roofline_block(res, args.traffic_from),
        }
        if world > 1:
            line["scalings"] = scalings
        if projection:
            line["strong_scaling_projection"] = projection
        if extras:
            line["configs"] = extras
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
