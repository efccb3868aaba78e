"""C5 (1024 chains, logistic n = 100,000, reflective kernel, thin 10): time per step with the library's in-kernel variates against
variates fed from a materialised stream (what the drawing, and the scratch reloads of its AS241 constants, cost a step)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fmcmc_amd import engine as E, _abi as abi
cfg = bench.Config("c5")
X, y, init = cfg.workload(cfg.chains, 0)
gm, gk = bench.device_objects(cfg, E, abi, X, y, "cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
init_d = torch.as_tensor(init).cuda()
logu = torch.empty((cfg.chains, iters), dtype=torch.float64, device="cuda")
z = torch.empty((cfg.chains, iters, gk.kz), dtype=torch.float64, device="cuda")
for fed in (False, True, False, True):
    best = 1e9
    for rep in range(3):
        st = E.ChainState(init_d, cfg.k)
        kw = {}
        if fed:
            E.rng_stream(st, gk, iters, seed=1215, chain_base=0, logu=logu, z=z)
            kw = dict(fed_logu=logu, fed_z=z)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = E.sweep(gm, gk, st, iters, thin=10, seed=1215, want_bits=False, check=False, **kw)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    print("C5 %s: %.2f us per step on %s, checksum %016x" % ("fed stream " if fed else "own variates", best, abi.last_kernel(),
          int(r.samples.view(torch.int64).sum().item()) & (2**64 - 1)), flush=True)
