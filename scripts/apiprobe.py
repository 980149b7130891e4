"""scripts/apiprobe.py -- the estimator surface away from the bench's call pattern: partial_fit per trajectory, transform of many
short trajectories, host-array transform / predict, libdistance on mid-size inputs.  Wall times; anomalies show as outliers."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA, KCenters, libdistance
warnings.simplefilter("ignore")
def timeit(tag, f, reps=3):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print("%-78s %9.2f ms" % (tag, 1e3 * min(ts)))
    return r
g = torch.Generator(device="cuda").manual_seed(2)
N, F = 2_000_000, 512
X = torch.randn(N, F, generator=g, device="cuda")
for T in (10000, 1000):
    seqs = list(X.view(N // T, T, F).unbind(0))
    m = timeit("tICA.fit 2M x 512 as %d x %d" % (N // T, T), lambda: tICA(n_components=10, lag_time=50).fit(seqs))
    def pf():
        mm = tICA(n_components=10, lag_time=50)
        for s in seqs:
            mm.partial_fit(s)
        return mm
    timeit("  the same through partial_fit per trajectory (%d calls)" % len(seqs), pf, reps=2)
    _ = m.eigenvalues_
    timeit("  transform (list of %d device trajectories)" % len(seqs), lambda: m.transform(seqs))
    timeit("  partial_transform per trajectory", lambda: [m.partial_transform(s) for s in seqs], reps=2)
host = [s.cpu().numpy() for s in list(X.view(200, 10000, F).unbind(0))[:100]]
m = tICA(n_components=10, lag_time=50).fit(host)
timeit("tICA.fit on 100 host trajectories (2 GB)", lambda: tICA(n_components=10, lag_time=50).fit(host))
Y = timeit("tICA.transform on 100 host trajectories (2 GB)", lambda: m.transform(host))
kc = timeit("KCenters(200).fit on the 1M x 10 host projection", lambda: KCenters(n_clusters=200, random_state=0).fit(Y))
timeit("KCenters.predict on it", lambda: kc.predict(Y))
del X
# libdistance
A = np.random.RandomState(0).randn(20000, 64).astype(np.float32)
B = np.random.RandomState(1).randn(5000, 64).astype(np.float32)
for metric in ("euclidean", "cityblock", "canberra"):
    timeit("libdistance.cdist 20000 x 5000 x 64 f32 %s (host in, host out: 400 MB of f64)" % metric, lambda: libdistance.cdist(A, B, metric), reps=2)
timeit("libdistance.pdist 20000 x 64 f32 euclidean (200M pairs -> 1.6 GB host)", lambda: libdistance.pdist(A, "euclidean"), reps=2)
timeit("libdistance.assign_nearest 20000 x 64 vs 5000 centres", lambda: libdistance.assign_nearest(A, B, "euclidean"))
timeit("libdistance.dist 20000 x 64 row 7", lambda: libdistance.dist(A, A[7], "euclidean"))
timeit("libdistance.sumdist 20000 x 64, 100000 pairs", lambda: libdistance.sumdist(A, "euclidean", np.random.RandomState(3).randint(0, 20000, (100000, 2)).astype(np.int64)), reps=2)
hostL = [np.random.RandomState(i).randn(10000, 512).astype(np.float32) for i in range(100)]
timeit("tICA.fit_transform on 100 host trajectories (2 GB)", lambda: tICA(n_components=10, lag_time=50).fit_transform(hostL))
timeit("tICA.fit + transform on the same (two passes over PCIe)", lambda: (lambda mm: mm.transform(hostL))(tICA(n_components=10, lag_time=50).fit(hostL)))
