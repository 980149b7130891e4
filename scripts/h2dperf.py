"""scripts/h2dperf.py -- PCIe-inclusive tICA fit from pageable numpy trajectories (bench.py's h2d_inclusive leg alone)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA
T, F, nh = 10000, 512, 200
rng = np.random.default_rng(0)
host = [rng.standard_normal((T, F), dtype=np.float32) for _ in range(nh)]
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    tICA(n_components=10, lag_time=100).fit(host[:20])
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        tICA(n_components=10, lag_time=100).fit(host)
        torch.cuda.synchronize(); t = time.perf_counter() - t
        print("%.1f GB/s  %.1f ms" % (nh * T * F * 4 / t / 1e9, 1e3 * t))
