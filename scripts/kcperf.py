"""Profiling helper: KCenters.fit on the bench's projection shape (10M x 10 f64, K = 200), with and without pruning."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from msmbuilder_amd import tICA, KCenters
n_seq, T, F = 1000, 10000, 512
X = bench.synth(torch, n_seq, T, F, 1234, torch.device("cuda"))
m = tICA(n_components=10, lag_time=100).fit(list(X.view(n_seq, T, F).unbind(0)))
Y = m.transform([X])[0]
del X
for name, Z in (("tICA projection", Y), ("white noise", torch.randn_like(Y))):
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        kc = KCenters(n_clusters=200, random_state=0).fit([Z])
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    import ctypes as _C
    from msmbuilder_amd import _lib as _L
    _st = (_C.c_int64 * 5)(); _L.check(_L.lib().msm_kcenters_last_stats(_st))
    print("   passes: %d plain + %d on the screen copy (%d B per row)" % (_st[1], _st[2], _st[4]))
    print("%s: KCenters(200).fit 10M x 10 f64: %.2f ms inertia %.6e ids[:4] %s" % (
        name, 1e3 * min(ts[1:]), kc.inertia_, kc.cluster_ids_[:4]))
# one rank's share of an 8-GPU run (1.25M rows): the single-process fit against the row-sharded library loop (a world of one)
Z = Y[:1_250_000].contiguous()
for name, env in (("single-process fit", None), ("sharded loop, world of one", "1")):
    KCenters._force_sharded = bool(env)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        kc = KCenters(n_clusters=200, random_state=0).fit([Z])
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    KCenters._force_sharded = False
    print("1.25M x 10 f64, %s: KCenters(200).fit %.2f ms  ids[:4] %s" % (name, 1e3 * min(ts[1:]), kc.cluster_ids_[:4]))
