"""convergence_gelman's device reduction at config C4's width: 512 chains x p = 50 x the second half (5,000 rows) of a
10,000-row history (1.02 GB read per pass): time of fmcmc_gelman_partial_dev (HIP events) and the achieved HBM rate."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmcmc_amd import _abi as abi
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 512
p = k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
S = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
row0, N = S // 2, S - S // 2
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn((Cn, k, S), dtype=torch.float64, device="cuda", generator=g) * 0.5 + 3.0
cols = torch.arange(p, dtype=torch.int32, device="cuda")
center = x[0, :, row0].contiguous()
L = abi.lib()
part = torch.zeros(int(L.fmcmc_gelman_partial_len(p)), dtype=torch.float64, device="cuda")
work = torch.empty(int(L.fmcmc_gelman_work_len(Cn, p)), dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run():
    rc = L.fmcmc_gelman_partial_dev(x.data_ptr(), Cn, k, S, row0, N, cols.data_ptr(), p, center.data_ptr(), work.data_ptr(),
                                    part.data_ptr(), C.c_void_p(st))
    assert rc == 0
run(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = min(ts)
byt = Cn * p * N * 8
print("gelman partial: %d chains x p=%d x N=%d rows (%.2f GB window): %.3f ms, %.0f GB/s of window bytes, %.1f GFLOP/s (syrk)" % (
    Cn, p, N, byt / 1e9, ms, byt / ms / 1e6, Cn * p * p * N * 2 / ms / 1e6))
# spot check against numpy on 8 chains
w = x[:8, :, row0:].cpu().numpy().transpose(0, 2, 1)
xb = w.mean(1) - center.cpu().numpy()
wk = work.cpu().numpy().reshape(Cn, p + p * p)
print("max |xbar err| %.2e, max rel |S_c err| %.2e" % (np.abs(wk[:8, :p] - xb).max(),
      max(np.abs(wk[c, p:].reshape(p, p) - np.cov(w[c].T, ddof=1)).max() / np.abs(np.cov(w[c].T, ddof=1)).max() for c in range(8))))
