"""scripts/shapeprobe.py -- tICA.fit on shapes the bench does not have: many short trajectories, ragged lengths, feature counts
that are not multiples of 128, lag 1.  Wall time of fit (fp32, device-resident rows) and the accumulation kernel's time."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from msmbuilder_amd import tICA, _lib
warnings.simplefilter("ignore")
def run(tag, seqs, lag, F):
    frames = sum(int(s.shape[0]) for s in seqs)
    for _ in range(2):
        m = tICA(n_components=10, lag_time=lag).fit(seqs)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        m = tICA(n_components=10, lag_time=lag).fit(seqs)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    km = bench.kernel_ms_of(m, _lib)
    print("%-44s fit %7.2f ms  kernel %7.2f ms  %6.1fM frames/s (fit)  kernel at the 10M x 512 rate would take %6.2f ms"
          % (tag, 1e3 * min(ts), km, frames / min(ts) / 1e6, frames * (F / 512.0) ** 2 * 50.1 / 1e7))
g = torch.Generator(device="cuda").manual_seed(1)
N = 4_000_000
X = torch.randn(N, 512, generator=g, device="cuda")
for T in (10000, 2500, 1000, 500, 250):
    run("4M x 512 as %d x %d, lag 100" % (N // T, T), list(X.view(N // T, T, 512).unbind(0)), 100, 512)
run("4M x 512 as 400 x 10000, lag 1", list(X.view(400, 10000, 512).unbind(0)), 1, 512)
# ragged: lengths 300 .. 6000
rs = np.random.RandomState(0)
lens, tot = [], 0
while tot < N - 6000:
    l = int(rs.randint(300, 6000)); lens.append(l); tot += l
offs = np.concatenate(([0], np.cumsum(lens)))
run("4M x 512 ragged (%d trajectories of 300..6000)" % len(lens), [X[offs[i]:offs[i + 1]] for i in range(len(lens))], 100, 512)
del X
for F in (500, 300, 171, 1000):
    n = 2_000_000
    Xf = torch.randn(n, F, generator=g, device="cuda")
    run("2M x %d as 200 x 10000, lag 100" % F, list(Xf.view(200, 10000, F).unbind(0)), 100, F)
    del Xf
