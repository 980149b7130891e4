"""Profiling helper: KCenters(200).fit on 10M x 10 float64 CLUSTERED rows (40 Gaussian blobs) -- the label-sorted fit
(tile summaries) against plain per-row pruning (MSM_KC_SORTED=0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmbuilder_amd import KCenters
g = torch.Generator(device="cuda").manual_seed(3)
n = 10_000_000
blobs = torch.randn(40, 10, generator=g, device="cuda", dtype=torch.float64) * 6.0
Z = blobs[torch.randint(0, 40, (n,), generator=g, device="cuda")] + 0.4 * torch.randn(n, 10, generator=g, device="cuda", dtype=torch.float64)
ts = []
for _ in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    kc = KCenters(n_clusters=200, random_state=0).fit([Z])
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
print("40 blobs: KCenters(200).fit 10M x 10 f64: %.2f ms (MSM_KC_SORTED=%s MSM_KC_SORT=%s) inertia %.9e ids[:4] %s" % (
    1e3 * min(ts[1:]), os.environ.get("MSM_KC_SORTED", "0"), os.environ.get("MSM_KC_SORT", "default"), kc.inertia_, kc.cluster_ids_[:4]))
