"""KCenters.predict-shaped assign_nearest (10M x 10 float64 rows, K = 200, device resident): screened kernel against the
exact one (MSM_ASSIGN_SCREEN=0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import libdistance as ld
n, m, K = 10_000_000, 10, 200
g = torch.Generator(device="cuda").manual_seed(0)
Z = torch.randn(n, 16, device="cuda", generator=g, dtype=torch.float64)
M = torch.randn(16, m, device="cuda", generator=g, dtype=torch.float64) / 4
X = (Z @ M).contiguous()
Y = X[torch.randint(0, n, (K,), device="cuda", generator=g)].cpu().numpy()
for env in ("1", "0", "1", "0"):
    os.environ["MSM_ASSIGN_SCREEN"] = env
    ld.assign_nearest(X, Y, "euclidean")
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        lab, inertia = ld.assign_nearest(X, Y, "euclidean")
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print("MSM_ASSIGN_SCREEN=%s  %.3f ms  inertia %.15e  labels[:5] %s" % (env, 1e3 * dt, inertia, lab[:5].tolist()))
