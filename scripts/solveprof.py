"""Profiling helper: where the time between tICA.fit and transform goes (export + finalise + eigensolve)."""
import cProfile, pstats, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA
torch.manual_seed(0)
n_seq, T, F = 200, 10000, 512
Z = torch.randn(n_seq * T, 16, device="cuda").cumsum(0) * 0.01
X = Z @ torch.randn(16, F, device="cuda") + torch.randn(n_seq * T, F, device="cuda")
seqs = list(X.view(n_seq, T, F).unbind(0))
warnings.simplefilter("ignore")
for _ in range(2):
    m = tICA(n_components=10, lag_time=100).fit(seqs); m.eigenvalues_
ts = []
pr = cProfile.Profile()
for _ in range(5):
    m = tICA(n_components=10, lag_time=100).fit(seqs)
    torch.cuda.synchronize()
    t = time.perf_counter(); pr.enable()
    ev = m.eigenvalues_
    pr.disable(); ts.append((time.perf_counter() - t) * 1e3)
print("solve (export + finalise + eigh): %s ms" % np.round(ts, 2))
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
