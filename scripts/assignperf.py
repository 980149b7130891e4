"""Profiling helper: exact assign_nearest at the two shapes VERDICT r1 names (KCenters.predict 10M x 10 f64 x K=200,
4M x 512 f32 x K=100), with the fp64-VALU bound next to each (3 separately rounded fp64 ops per pair-element for f64 input;
fp32 sub + cvt + fp64 mul + add for f32: 12 / 14 issue cycles per wave-level pair-element, 1024 SIMDs at 2.4 GHz)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import libdistance
def run(n, m, K, dt, cyc):
    X = torch.randn(n, m, device="cuda", dtype=dt)
    Y = X[torch.randperm(n, device="cuda")[:K]].cpu().numpy()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        lab, inertia = libdistance.assign_nearest(X, Y, "euclidean")
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    t = min(ts[1:])
    bound = n * m * K / 64 * cyc / (1024 * 2.4e9)
    print("assign_nearest %dx%d %s K=%d: %.2f ms  (fp64-VALU bound %.2f ms -> %.2f of it; %.2fT pair-elements/s)" % (
        n, m, str(dt).replace("torch.", ""), K, 1e3 * t, 1e3 * bound, bound / t, n * m * K / t / 1e12))
run(10_000_000, 10, 200, torch.float64, 12)
run(4_000_000, 512, 100, torch.float32, 14)
run(4_000_000, 512, 8, torch.float32, 14)
run(2_000_000, 256, 100, torch.float64, 12)
run(10_000_000, 10, 200, torch.float32, 14)
