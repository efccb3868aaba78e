"""C4 with the owners' Student-t variates drawn in-kernel (default) against a stream materialised in front of the sweep
(fmcmc_rng_stream_dev + rng_mode FED): what the drawing costs the evaluator waves that share a SIMD with an owner."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fmcmc_amd import engine as E, _abi as abi
cfg = bench.Config("c4")
X, y, init = cfg.workload(512, 0)
gm, gk = bench.device_objects(cfg, E, abi, X, y, torch.device("cuda", 0))
nst = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for mode in ("philox", "fed", "philox", "fed"):
    best = 1e9
    for _ in range(3):
        st = E.ChainState(init, 50)
        kw = {}
        torch.cuda.synchronize(); t = time.time()
        if mode == "fed":
            logu, z = E.rng_stream(st, gk, nst, seed=1215)
            kw = dict(fed_logu=logu, fed_z=z)
        r = E.sweep(gm, gk, st, nst, seed=1215, want_bits=False, want_draws=False, check=False, **kw)
        torch.cuda.synchronize(); best = min(best, time.time() - t)
    print("%-7s %.2f us per step (incl. the stream kernel), kernel %s, checksum %016x" % (mode, best / nst * 1e6, abi.last_kernel(), int(r.samples.view(torch.int64).sum().item()) & (2**64 - 1)))
