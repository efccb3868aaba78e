import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import synth_linreg
from oracle import oracle as O
from fmcmc_amd import engine as E
C, n, p, nsteps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
X, y = synth_linreg(n, p, 11 + n)
rng = np.random.default_rng(5)
init = np.asarray([0.0] * (p + 1) + [float(np.std(y))])[None, :] + 0.1 * rng.standard_normal((C, p + 2))
k = p + 2
ok = O.Kernel(O.K_NORMAL, k, scale=0.05)
ro = O.run(O.Model(O.FAM_LINREG, X, y), ok, init, nsteps=nsteps, seed=1215)
gk = E.KernelSpec(O.K_NORMAL, k, ok.mu, ok.scale, ok.lb, ok.ub, ok.fixed)
st = E.ChainState(init, k)
rg = E.sweep(E.DeviceModel(1, X, y), gk, st, nsteps, seed=1215, check=False)
torch.cuda.synchronize()
gs, os_ = rg.samples.cpu().numpy(), ro.samples_cks
gd, od = rg.draws.cpu().numpy(), ro.draws_cks
gl, ol = rg.logpost.cpu().numpy(), ro.logpost
for c in range(C):
    bad_d = np.nonzero((gd[c] != od[c]).any(0))[0]
    bad_l = np.nonzero(gl[c] != ol[c])[0]
    bad_s = np.nonzero((gs[c] != os_[c]).any(0))[0]
    print("chain", c, "first bad draw row", bad_d[:3], "logpost", bad_l[:3], "sample", bad_s[:3])
    if bad_l.size:
        r = bad_l[0]
        print("  row", r, "gpu lp %.17g oracle %.17g" % (gl[c, r], ol[c, r]), "draw gpu", gd[c, :, r], "oracle", od[c, :, r])
print("status", rg.status.cpu().numpy(), ro.status)
