"""Wall time of back-to-back tICA.fit calls with and without the host eigensolve in between."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmbuilder_amd import tICA
F, T, n_seq = 512, 10000, 1000
X = torch.randn(n_seq * T, F, device="cuda")
seqs = list(X.view(n_seq, T, F).unbind(0))
warnings.simplefilter("ignore")
def one(solve, sleep=0.0):
    t = time.perf_counter(); m = tICA(n_components=10, lag_time=100).fit(seqs); torch.cuda.synchronize(); t1 = time.perf_counter()
    if solve: m.eigenvalues_
    t2 = time.perf_counter()
    if sleep: time.sleep(sleep)
    return 1e3 * (t1 - t), 1e3 * (t2 - t1)
for label, solve, sleep in (("fit only", False, 0), ("fit+solve", True, 0), ("fit+solve+50ms sleep", True, 0.05), ("fit only", False, 0)):
    one(solve, sleep)
    r = [one(solve, sleep) for _ in range(4)]
    print("%-22s fit %s  solve %s" % (label, " ".join("%.1f" % a for a, _ in r), " ".join("%.1f" % b for _, b in r)))
