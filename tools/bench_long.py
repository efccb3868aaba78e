"""Time per step of C2's shape against the call length (step windows of the stream-fed kernels): tools/bench_long.py [chains]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fmcmc_amd import engine as E, _abi as abi
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
k = 5
X, y, init = bench.Config("c2").workload(chains, 0)
gm = E.DeviceModel(abi.FAM_GAUSSIAN_LINREG, X, y)
gk = E.KernelSpec(abi.KERNEL_NORMAL, k, np.zeros(k), np.full(k, 0.02), np.full(k, -E.DBL_MAX), np.full(k, E.DBL_MAX), np.zeros(k, np.uint8))
for nsteps in (10000, 30000, 120000, 120000):
    st = E.ChainState(init, k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t = time.time(); e0.record()
    r = E.sweep(gm, gk, st, nsteps, seed=1215, want_bits=False, check=False)
    e1.record(); torch.cuda.synchronize()
    print("nsteps %6d: %.3f us per step (events), %.3f (wall), kernel %s, window knob %s" % (
        nsteps, e0.elapsed_time(e1) * 1e3 / nsteps, (time.time() - t) * 1e6 / nsteps, abi.last_kernel(), os.environ.get("FMCMC_AMD_DEBUG", "-")))
    del r
