#!/usr/bin/env python3
"""Headline benchmark: MH samples/s (chains x iterations / s), BASELINE.json configs[1]:
1024 chains x 5-parameter Gaussian linear regression, n = 10,000, kernel_normal, 1 MI355X.

One "step" = one full sweep of the hot path over the workload: all chains of the rank advance
`--iters` MH iterations (default 10,000 = the config's nsteps) inside ONE fused kernel launch,
writing ans, logpost and draws exactly like the reference (R/mcmc.R:728-734,822-823).
Inputs (X, y, initial states) are resident in HBM before the timed region.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (weak scaling:
  every rank runs its own 1024-chain shard, chain ids continue across ranks, no data-path collective)
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

N_OBS, P_COV, K_PAR = 10000, 3, 5
CHAINS_PER_GPU = 1024
SCALE = 0.02
DATA_SEED, CHAIN_SEED = 20260102, 1215


def make_workload(chains, chain_base):
    rng = np.random.default_rng(DATA_SEED)
    X = rng.standard_normal((N_OBS, P_COV))
    beta = np.array([3.0, 2.0, -1.0, 0.5])
    y = beta[0] + X @ beta[1:] + 4.0 * rng.standard_normal(N_OBS)
    irng = np.random.default_rng(DATA_SEED + 1)
    jit = 0.1 * irng.standard_normal((chain_base + chains, K_PAR))[chain_base:]
    init = np.array([0.0, 0.0, 0.0, 0.0, float(np.std(y, ddof=1))])[None, :] + jit
    init[:, -1] = np.abs(init[:, -1])
    return X, y, np.ascontiguousarray(init)


def flops_per_sample():
    # E * n * (2(p-1)+3), p-1 = 3 covariates: 3 fma + 1 sub + 1 fma per observation (the MFMA path
    # executes 4 fma + 1 fma = 10 flop per observation; the ALGORITHMIC 9 is what achieved/peak is computed from)
    return N_OBS * (2 * P_COV + 3)


def cpu_baseline(seconds_budget=24.0):
    """The oracle (PHILOX/canonical mode = same outputs as the GPU) timed on the host cores, on a
    bounded sample of the SAME workload: `m` chains x 10,000 iterations per host thread."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    iters = 10000
    # calibrate with ALL host threads busy (SMT and memory contention included), then size the sample
    X, y, init0 = make_workload(cores, 0)
    m = O.Model(O.FAM_LINREG, X, y)
    k = O.Kernel(O.K_NORMAL, K_PAR, scale=SCALE)

    def cal(tix):
        O.run(m, k, init0[tix:tix + 1], nsteps=201, seed=CHAIN_SEED, chain_base=tix, want_draws=True)

    t = time.time()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(cal, range(cores)))
    per = (time.time() - t) / 200            # seconds per iteration of one chain per thread, under load
    per_thread = int(max(1, min(64, seconds_budget / (per * iters))))
    chains = cores * per_thread
    _, _, init = make_workload(chains, 0)

    def one(tix):  # ctypes releases the GIL: one oracle call per host thread
        lo = tix * per_thread
        O.run(m, k, init[lo:lo + per_thread], nsteps=iters, seed=CHAIN_SEED, chain_base=lo, want_draws=True)

    t = time.time()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(one, range(cores)))
    dt = time.time() - t
    return {"value": chains * (iters - 1) / dt, "unit": "MH samples/s", "cores": cores, "kind": "port",
            "seconds": dt,
            "sample": "%d chains x %d iterations of the same workload (%d chains per host thread), "
                      "oracle/fmcmc_oracle.c in PHILOX/canonical mode: a C restatement of fmcmc's R loop that "
                      "produces the GPU's exact outputs; optimistic vs interpreted R" % (chains, iters, per_thread)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--iters", type=int, default=10000, help="MH iterations per sweep (config nsteps)")
    ap.add_argument("--chains", type=int, default=CHAINS_PER_GPU, help="chains per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from fmcmc_amd import engine as E, _abi as abi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    chains, iters = args.chains, args.iters
    chain_base = rank * chains
    X, y, init = make_workload(chains, chain_base)
    gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y, device=dev)
    gk = E.KernelSpec(abi.KERNEL_NORMAL, K_PAR, np.zeros(K_PAR), np.full(K_PAR, SCALE),
                      np.full(K_PAR, -E.DBL_MAX), np.full(K_PAR, E.DBL_MAX), np.zeros(K_PAR, np.uint8), device=dev)
    init_d = torch.as_tensor(init).to(dev)

    # the canonical Philox stream of a sweep is generated by its own small kernel into reusable HBM buffers,
    # then the sweep consumes it (bit-identical to letting the library do both); two launches per step
    logu_buf = torch.empty((chains, iters), dtype=torch.float64, device=dev)
    z_buf = torch.empty((chains, iters, K_PAR), dtype=torch.float64, device=dev)
    ev_mid = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    cur = {"s": None}

    def one_step():
        st = E.ChainState(init_d, K_PAR, device=dev)
        E.rng_stream(st, gk, iters, seed=CHAIN_SEED, chain_base=chain_base, logu=logu_buf, z=z_buf)
        if cur["s"] is not None:
            ev_mid[cur["s"]].record()
        return E.sweep(gm, gk, st, iters, seed=CHAIN_SEED, chain_base=chain_base, want_logpost=True,
                       want_draws=True, want_bits=False, check=False, fed_logu=logu_buf, fed_z=z_buf)

    for _ in range(args.warmup):
        out = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for s in range(args.steps):
        cur["s"] = s
        ev[s][0].record()            # same stream the kernels are launched on
        out = one_step()             # rng_fill_kernel | ev_mid | output allocation + mh_sweep kernel
        ev[s][1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert int(out.status.abs().sum().item()) == 0, "a chain reported an error"
    # kernel duration: HIP events around the launch (the bracket also contains the output allocation
    # memset, < 1% of the sweep); rocprofv3 --kernel-trace --stats of the same command: profiles/
    kern_ms = float(np.mean([ev_mid[s].elapsed_time(ev[s][1]) for s in range(args.steps)]))   # sweep kernel (+ its output memsets)
    rng_ms = float(np.mean([ev[s][0].elapsed_time(ev_mid[s]) for s in range(args.steps)]))
    samples_per_step = chains * (iters - 1)
    value = world * samples_per_step * args.steps / elapsed
    acc = float(out.accept_count.double().mean().item()) / (iters - 1)

    if rank == 0:
        ach_tflops = samples_per_step * flops_per_sample() / (kern_ms * 1e-3) / 1e12
        out_bytes = chains * iters * (2 * K_PAR + 1) * 8
        traffic = None
        pj = os.path.join(ROOT, "profiles", "latest_pmc.json")
        if os.path.exists(pj):
            try:
                traffic = json.load(open(pj)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "MH samples/sec (chains x iters / s), 1024 chains, 5-param linreg n=10k",
            "value": value, "unit": "MH samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: %d chains/GPU x %d-param Gaussian linreg n=%d, kernel_normal(scale=%g), "
                                   "nsteps=%d, outputs ans+logpost+draws" % (chains, K_PAR, N_OBS, SCALE, iters),
                       "chains_per_gpu": chains, "iters_per_step": iters, "accept_rate": acc,
                       "parallelism": "chains sharded, %d rank(s), no data-path collective" % world},
            "roofline": {"bound": "mfma", "pipe": "fp64 MFMA (v_mfma_f64_4x4x4: x.beta - y) + fp64 VALU (r^2 accumulate); MI355X fp64 matrix peak == fp64 vector peak",
                         "achieved": ach_tflops, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tflops / PEAK_FP64_TFLOPS, "traffic": traffic,
                         "kernel": "mh_sweep_mfma<1,1,20,false>", "kernel_ms": kern_ms, "rng_fill_kernel_ms": rng_ms,
                         "flops_per_sample": flops_per_sample(),
                         "hbm": {"achieved_GBps": out_bytes / (kern_ms * 1e-3) / 1e9, "peak_GBps": PEAK_HBM_GBS,
                                 "algorithmic_bytes_per_launch": out_bytes}},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
