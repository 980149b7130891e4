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
    print("%s: KCenters(200).fit 10M x 10 f64: %.2f ms (MSM_KC_PRUNE=%s) inertia %.6e ids[:4] %s" % (
        name, 1e3 * min(ts[1:]), os.environ.get("MSM_KC_PRUNE", "1"), kc.inertia_, kc.cluster_ids_[:4]))
