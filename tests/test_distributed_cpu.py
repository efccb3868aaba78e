"""The N > 1 path on CPU: two gloo ranks own contiguous chain blocks (chain ids are global, so a chain's
stream does not depend on the sharding), build their Gelman partials, join them with ONE all-reduce(sum)
and every rank derives the same R-hat as an unsharded run.  On the GPU the partial comes from
fmcmc_gelman_partial_dev and the all-reduce is RCCL; the host half exercised here is identical."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fmcmc_amd import _abi as abi, shard_bounds
    from oracle import oracle as O
    from conftest import synth_linreg
    from test_abi import numpy_gelman_partial
    Cn, nsteps, k = 6, 400, 4
    X, y = synth_linreg(300, 2, 77)
    init = np.array([0, 0, 0, 4.0])[None, :] + 0.2 * np.random.default_rng(1).standard_normal((Cn, k))
    lo, hi = shard_bounds(Cn, world, rank)
    kern = O.Kernel(O.K_NORMAL, k, scale=0.1)
    # this rank's shard: same global chain ids as the unsharded run
    r = O.run(O.Model(O.FAM_LINREG, X, y), kern, init[lo:hi], nsteps=nsteps, seed=5, chain_base=lo)
    window = r.samples[:, nsteps // 2:, :]
    N = window.shape[1]
    center = torch.zeros(k, dtype=torch.float64)
    if rank == 0:
        center.copy_(torch.as_tensor(window[0, 0]))
    dist.all_reduce(center)                               # same centre on every rank
    part = torch.as_tensor(numpy_gelman_partial(window, center.numpy()))
    dist.all_reduce(part)                                 # the engine's only collective
    psrf = np.empty(k); mps = C.c_double()
    dp = C.POINTER(C.c_double)
    pn = np.ascontiguousarray(part.numpy())
    rc = abi.lib().fmcmc_gelman_finish(pn.ctypes.data_as(dp), k, N, psrf.ctypes.data_as(dp), C.byref(mps))
    q.put((rank, rc, mps.value, psrf.tolist(), r.samples[:, -1, :].tolist(), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gelman_all_reduce_equals_unsharded(O):
    from conftest import synth_linreg
    world, port = 2, 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # unsharded reference on one process
    Cn, nsteps, k = 6, 400, 4
    X, y = synth_linreg(300, 2, 77)
    init = np.array([0, 0, 0, 4.0])[None, :] + 0.2 * np.random.default_rng(1).standard_normal((Cn, k))
    full = O.run(O.Model(O.FAM_LINREG, X, y), O.Kernel(O.K_NORMAL, k, scale=0.1), init, nsteps=nsteps, seed=5)
    opsrf, ompsrf = O.gelman(full.samples[:, nsteps // 2:, :])
    last = np.concatenate([np.array(r[4]) for r in res])
    assert np.array_equal(last, full.samples[:, -1, :])          # sharding never changes a chain
    assert [r[5] for r in res] == [(0, 3), (3, 6)]
    for rank, rc, mpsrf, psrf, _, _ in res:
        assert rc == 0
        assert abs(mpsrf - ompsrf) < 1e-9 * ompsrf and np.allclose(psrf, opsrf, rtol=1e-9)
    assert res[0][2] == res[1][2]                                  # every rank decides identically


def _bench_cli(args, env_drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"), env_extra=None):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    env.update(env_extra or {})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=300)


def test_bench_refuses_a_launcher_whose_world_differs_from_gpus():
    """`--gpus N` is a contract, not a label: under a launcher the rank count must match (no GPU needed: checked first)."""
    r = _bench_cli(["--gpus", "4"], env_extra=dict(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "--gpus 4 but the launcher started WORLD_SIZE=2" in r.stderr
    r = _bench_cli(["--gpus", "0"])
    assert r.returncode != 0 and "--gpus must be >= 1" in r.stderr


def test_bench_gpus_n_without_a_launcher_starts_n_rank_processes():
    """No rank variables in the environment: bench.py itself starts N fresh ranks (R/mcmc.R:536-545 creates its own
    workers).  On this GPU-less box every rank ends with the engine's 'needs an MI355X' refusal -- N of them, and the
    parent reports the ranks' return codes and fails."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-side check of the launcher (the GPU suite runs the real thing)")
    r = _bench_cli(["--gpus", "3", "--backend", "gloo", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs an MI355X") == 3
    assert "rank return codes [1, 1, 1]" in r.stderr
