"""Diagnostic: per-wave cycle shares of mh_sweep_spec at config C3 (kernel_adapt, 1024 chains; FMCMC_AMD_DEBUG=mode=8).  Not a benchmark.
   python tools/stamp_c3.py [chains=1024] [c3|c2]"""
import os, sys
os.environ["FMCMC_AMD_DEBUG"] = os.environ.get("FMCMC_AMD_DEBUG", "") + (",mode=8" if os.environ.get("FMCMC_AMD_DEBUG") else "mode=8")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fmcmc_amd import engine as E, _abi as abi
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = bench.Config(sys.argv[2] if len(sys.argv) > 2 else "c3")
X, y, init = cfg.workload(C, 0)
gm, gk = bench.device_objects(cfg, E, abi, X, y, "cuda:0")
nsteps = 3000
st = E.ChainState(torch.as_tensor(init).cuda(), cfg.k)
r = E.sweep(gm, gk, st, nsteps, seed=1215, want_draws=True, want_bits=False, check=False)
torch.cuda.synchronize()
kn = abi.last_kernel().split(":")[-1]
cw = int(kn[-1]) if kn.startswith("spec-lat") else 4
nb = (C + cw - 1) // cw
d = r.draws.reshape(-1)[: nb * 12 * 4].cpu().numpy().reshape(nb, 12, 4)
per = d[:, :, :3] / np.maximum(d[:, :, 3:4], 1)
print("kernel %s: ticks per MH step (s_memtime), median over %d workgroups" % (kn, nb))
print("waves 0-7 compute = (flag wait, eval, -), waves 8-11 owners = (flag wait, phase to publish, stores)")
med = np.median(per, axis=0)
for w in range(12):
    print("wave %2d: %8.0f %8.0f %8.0f | %8.0f" % (w, med[w, 0], med[w, 1], med[w, 2], med[w].sum()))
