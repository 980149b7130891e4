"""One rank's share of the bench at N = 8 (125 of 1000 trajectories): the row-sharded k-centers loop in a world of one on the
share's projection (1.25M x 10 float64, K = 200), wall time per fit and the loop's counters -- for rocprofv3 --kernel-trace
--stats (per-kernel durations of the round's kernels)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from msmbuilder_amd import tICA, KCenters, _lib
n_seq, T, F = 125, 10000, 512
X = bench.synth(torch, n_seq, T, F, 1234, torch.device("cuda"))
m = tICA(n_components=10, lag_time=100).fit(list(X.view(n_seq, T, F).unbind(0)))
Y = m.transform([X])[0]
del X
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for name, forced in (("single-process fit", False), ("sharded loop, world of one", True)):
    KCenters._force_sharded = forced
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        kc = KCenters(n_clusters=200, random_state=0).fit([Y])
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    st = (C.c_int64 * 5)(); _lib.check(_lib.lib().msm_kcenters_last_stats(st))
    print("1.25M x 10 f64, %-28s KCenters(200).fit min %.3f ms median %.3f ms  passes: %d plain + %d batched  ids[:4] %s inertia %.9e" % (
        name, 1e3 * min(ts[1:]), 1e3 * float(np.median(ts[1:])), st[1], st[2], kc.cluster_ids_[:4], kc.inertia_))
KCenters._force_sharded = False
