"""scripts/fitprof.py -- host-side profile of tICA.fit on 1,000 device-resident trajectories (the bench step's fit):
which Python / ctypes calls sit in front of the MFMA kernel."""
import cProfile, os, pstats, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA
n, T, F = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 10000, 512
X = torch.randn(n * T, F, device="cuda")
seqs = list(X.view(n, T, F).unbind(0))
warnings.simplefilter("ignore")
for _ in range(3):
    m = tICA(n_components=10, lag_time=100).fit(seqs)
torch.cuda.synchronize()
import msmbuilder_amd._lib as _lib
L = _lib.lib()
orig = {}
acc = {}
def wrap(name):
    f = getattr(L, name)
    def g(*a):
        t = time.perf_counter(); r = f(*a); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
    return f, g
pr = cProfile.Profile()
reps = 10
t0 = time.perf_counter()
pr.enable()
for _ in range(reps):
    m = tICA(n_components=10, lag_time=100).fit(seqs)
pr.disable()
t_host = (time.perf_counter() - t0) / reps
torch.cuda.synchronize()
print("host time per fit call (returns before the kernel ends? %.3f ms)" % (1e3 * t_host))
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
