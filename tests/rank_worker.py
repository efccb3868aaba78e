"""One rank of the multi-process GPU tests (tests/test_gpu_multirank.py): a FRESH python process that joins a gloo group,
runs the sharded MCMC() of the product on the one visible GPU and writes what it got to an .npz.  Not a test module."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, out, case = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import fmcmc_amd as f
    from conftest import synth_linreg
    X, y = synth_linreg(1200, 2, 77)
    fun = f.gaussian_linreg(X, y)
    res = {}
    if case == "gelman":
        nch = 6
        init = np.array([0, 0, 0, 4.0])[None, :] + 0.2 * np.random.default_rng(1).standard_normal((nch, 4))
        chk = f.convergence_gelman(200, threshold=1.03)
        dc = f.MCMC(init, fun, 4000, seed=5, nchains=nch, kernel=f.kernel_normal(scale=0.05), conv_checker=chk,
                    _return_device=True)
        res = dict(samples=dc.samples.cpu().numpy(), iters=dc.iters, hist_end=[h[0] for h in chk.history],
                   hist_val=[h[1] for h in chk.history], converged=int(dc.converged), chain_base=dc.chain_base)
    elif case == "ram_gelman":
        nch = 5                                   # uneven split: 2 + 3 chains
        init = np.array([0, 0, 0, 4.0])[None, :] + 0.2 * np.random.default_rng(2).standard_normal((nch, 4))
        chk = f.convergence_gelman(300, threshold=1.02)
        kr = f.kernel_ram()
        dc = f.MCMC(init, fun, 3000, seed=11, nchains=nch, burnin=30, thin=2, kernel=kr, conv_checker=chk, _return_device=True)
        res = dict(samples=dc.samples.cpu().numpy(), iters=dc.iters, hist_end=[h[0] for h in chk.history],
                   hist_val=[h[1] for h in chk.history], converged=int(dc.converged), chain_base=dc.chain_base,
                   Sigma=kr._state.Sigma.cpu().numpy() if kr._state is not None else np.zeros((0, 4, 4)))
    elif case == "fewer":                         # nchains < world: the last rank holds no chain at all
        dc = f.MCMC([0, 0, 0, 4.0], fun, 500, seed=3, nchains=1, kernel=f.kernel_normal(scale=0.05), _return_device=True)
        res = dict(samples=dc.samples.cpu().numpy(), iters=dc.iters, chain_base=dc.chain_base)
    elif case == "seed_none":
        dc = f.MCMC(np.tile([0, 0, 0, 4.0], (4, 1)), fun, 300, nchains=4, kernel=f.kernel_normal(scale=0.05), _return_device=True)
        res = dict(samples=dc.samples.cpu().numpy(), seed=np.array([f.get_seed()], dtype=np.int64), chain_base=dc.chain_base)
    np.savez(out, **{k: np.asarray(v) for k, v in res.items()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
