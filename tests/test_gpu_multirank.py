"""The N > 1 path of the PRODUCT on hardware: MCMC() with torch.distributed initialised -- shard_bounds, global chain ids,
DeviceChains.nchains_total, the seed broadcast, convergence_gelman.check_device's two all-reduces -- run by two fresh
processes on the one GPU of the box (gloo moves the CUDA tensors of the all-reduce through the host; on a multi-GPU node the
same code runs one rank per GPU over RCCL).  Replaces, for all chains, the PSOCK fan-out of R/mcmc.R:536-641 and the
checker loop of R/mcmc.R:841-1019."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _run_ranks(tmp_path, case, world=2):
    port = 29600 + (os.getpid() * 7 + abs(hash(case))) % 1500
    outs = [str(tmp_path / ("%s_rank%d.npz" % (case, r))) for r in range(world)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rank_worker.py"), str(r), str(world), str(port),
                               outs[r], case], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    for p, o in zip(procs, logs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(o) for o in outs]


@pytest.fixture(scope="module")
def setup():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import fmcmc_amd as f
    from conftest import synth_linreg
    X, y = synth_linreg(1200, 2, 77)
    return f, f.gaussian_linreg(X, y)


def test_two_ranks_autostop_equals_the_unsharded_run(setup, tmp_path):
    f, fun = setup
    nch = 6
    init = np.array([0, 0, 0, 4.0])[None, :] + 0.2 * np.random.default_rng(1).standard_normal((nch, 4))
    chk = f.convergence_gelman(200, threshold=1.03)
    full = f.MCMC(init, fun, 4000, seed=5, nchains=nch, kernel=f.kernel_normal(scale=0.05), conv_checker=chk)
    assert len(chk.history) >= 3                                     # several bulks, several all-reduces
    r = _run_ranks(tmp_path, "gelman")
    assert [int(x["chain_base"]) for x in r] == [0, 3]
    both = np.concatenate([x["samples"] for x in r])                 # [C][k][S]
    assert np.array_equal(_bits(both.transpose(0, 2, 1)), _bits(full.as_array()))   # sharding never changes a chain
    for x in r:
        assert list(x["hist_end"]) == [h[0] for h in chk.history]   # the same bulks, the same stop
        assert np.allclose(x["hist_val"], [h[1] for h in chk.history], rtol=1e-9)
        assert list(x["iters"]) == list(full.iters) and int(x["converged"]) == 1
    assert np.array_equal(_bits(r[0]["hist_val"]), _bits(r[1]["hist_val"]))          # every rank decides identically


def test_two_ranks_uneven_split_ram_with_burnin_and_thinning(setup, tmp_path):
    f, fun = setup
    nch = 5
    init = np.array([0, 0, 0, 4.0])[None, :] + 0.2 * np.random.default_rng(2).standard_normal((nch, 4))
    chk = f.convergence_gelman(300, threshold=1.02)
    kr = f.kernel_ram()
    full = f.MCMC(init, fun, 3000, seed=11, nchains=nch, burnin=30, thin=2, kernel=kr, conv_checker=chk)
    r = _run_ranks(tmp_path, "ram_gelman")
    assert [int(x["chain_base"]) for x in r] == [0, 2] and [x["samples"].shape[0] for x in r] == [2, 3]
    both = np.concatenate([x["samples"] for x in r])
    assert np.array_equal(_bits(both.transpose(0, 2, 1)), _bits(full.as_array()))
    assert np.array_equal(_bits(np.concatenate([x["Sigma"] for x in r])), _bits(kr._state.Sigma.cpu().numpy()))
    for x in r:
        assert list(x["hist_end"]) == [h[0] for h in chk.history]
        assert np.allclose(x["hist_val"], [h[1] for h in chk.history], rtol=1e-9)


def test_fewer_chains_than_ranks(setup, tmp_path):
    f, fun = setup
    full = f.MCMC([0, 0, 0, 4.0], fun, 500, seed=3, nchains=1, kernel=f.kernel_normal(scale=0.05))
    r = _run_ranks(tmp_path, "fewer")
    shapes = [x["samples"].shape for x in r]
    assert sorted(s[0] for s in shapes) == [0, 1]                    # one rank runs the chain, the other none
    got = [x["samples"] for x in r if x["samples"].shape[0] == 1][0]
    assert np.array_equal(_bits(got[0].T), _bits(full.data))


def test_seed_none_is_rank_zeros_seed_everywhere(setup, tmp_path):
    f, fun = setup
    r = _run_ranks(tmp_path, "seed_none")
    seed = int(r[0]["seed"][0])
    assert int(r[1]["seed"][0]) == seed                              # broadcast from rank 0 (R/mcmc.R:455-456 sets ONE seed)
    full = f.MCMC(np.tile([0, 0, 0, 4.0], (4, 1)), fun, 300, seed=seed, nchains=4, kernel=f.kernel_normal(scale=0.05))
    both = np.concatenate([x["samples"] for x in r])
    assert np.array_equal(_bits(both.transpose(0, 2, 1)), _bits(full.as_array()))   # MCMC_OUTPUT's seed reproduces the run


def _bench(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_extra or {})
    return subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True)


def test_bench_multi_rank_path_runs_and_matches_the_single_rank_run():
    """bench.py's N > 1 path executed: two FRESH processes (started before anything touches the GPU; both on the one GPU of
    the box, gloo instead of RCCL) run `--gpus 2 --config c4` with 64 chains per rank -- process group, barriers, the MAX
    all-reduce of the elapsed time, chain_base = rank x chains, and C4's Gelman check with its all-reduce INSIDE the timed
    steps.  The R-hat the line reports is the one of a single process running all 128 chains: sharding changes nothing
    (replaces the PSOCK fan-out of R/mcmc.R:536-641)."""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    common = ["--config", "c4", "--chains", "64", "--iters", "2000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    port = 29100 + os.getpid() % 800
    procs = [_bench(["--gpus", "2", "--backend", "gloo"] + common,
                    dict(RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    line = json.loads([l for l in outs[0][0].strip().splitlines() if l.startswith("{")][-1])
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]          # rank 0 alone prints the line
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["backend"] == "gloo"
    assert line["config"]["chain_base_of_rank"] == [0, 64]
    # 2 steps x 2 bulks of 1000 iterations, each followed by one check = two all-reduces (centre + partial sums)
    assert line["config"]["all_reduce_calls_in_timed_steps"] == 2 * 2 * 2
    assert line["value"] > 0 and line["roofline"]["bulks_per_step"] == 2
    single = _bench(["--gpus", "1", "--config", "c4", "--chains", "128", "--iters", "2000", "--steps", "2", "--warmup", "1",
                     "--no-cpu-baseline"])
    o, e = single.communicate(timeout=600)
    assert single.returncode == 0, e[-3000:]
    one = json.loads([l for l in o.strip().splitlines() if l.startswith("{")][-1])
    assert one["n_gpus"] == 1 and one["config"]["chain_base_of_rank"] == [0]
    # the same 128 chains, bit for bit; their R-hat to the last ulps (the sum over the chains is rank 0's + rank 1's partial
    # instead of one fixed-order sum over 128)
    assert abs(line["roofline"]["rhat_last"] - one["roofline"]["rhat_last"]) <= 1e-13 * one["roofline"]["rhat_last"]


def test_bench_gpus_2_with_no_rank_variables_launches_two_ranks_itself():
    """The shape of the driver's command when no launcher is around it: `python bench.py --gpus 2 ...` with NO rank
    variables in the environment.  bench.py starts two fresh rank processes itself (before it touches the GPU), relays
    rank 0's line, and the line proves two ranks were there: n_gpus, chain_base_of_rank, ranks_seen as gathered through the
    process group.  (R/mcmc.R:536-545: the reference creates its own workers.)"""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", "c4",
                        "--chains", "64", "--iters", "2000", "--steps", "2", "--no-cpu-baseline"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["world"] == 2
    assert line["config"]["chain_base_of_rank"] == [0, 64]
    seen = line["config"]["ranks_seen"]
    assert [x["rank"] for x in seen] == [0, 1] and len({x["pid"] for x in seen}) == 2
    assert all(x["arch"] == "gfx950" for x in seen)
    assert "bench.py itself" in line["config"]["launched_by"]
    assert line["config"]["all_reduce_calls_in_timed_steps"] == 2 * 2 * 2


def test_bench_strong_scaling_divides_the_headline_chains():
    """--scaling strong: C2's 1024 chains divided over the ranks (512 each), chain ids continuing; the line says which
    scaling ran.  A launcher world that differs from --gpus is refused."""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")
    env = {k: v for k, v in os.environ.items() if k not in drop}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--scaling", "strong",
                        "--iters", "2000", "--steps", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
    assert line["config"]["chains_per_gpu"] == 512 and line["config"]["chain_base_of_rank"] == [0, 512]
    # both scalings in the one invocation: the strong leg on the latency form (two chains per workgroup), the weak leg on the
    # kernel of full launches with 2 x 1024 chains in the job
    sc = line["scalings"]
    assert sc["strong"]["chains_per_gpu"] == 512 and sc["strong"]["total_chains"] == 1024 and sc["strong"]["kernel"] == ["lat2"]
    assert sc["weak"]["chains_per_gpu"] == 1024 and sc["weak"]["total_chains"] == 2048 and sc["weak"]["kernel"] == ["mfma"]
    assert sc["strong"]["value"] == line["value"] and sc["weak"]["value"] > 0
    assert "weak scaling" not in line["metric"]
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "but the launcher started WORLD_SIZE=2" in bad.stderr
