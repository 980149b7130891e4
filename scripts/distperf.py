"""Profiling helper: exact-arithmetic distance kernels on long rows (HBM-bound scans and VALU-bound assigns)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import KCenters, libdistance as ld

def timeit(fn, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) * 1e3)
    return best

for dtype, N, F in ((torch.float32, 4_000_000, 512), (torch.float64, 2_000_000, 256), (torch.float32, 4_000_000, 64)):
    X = torch.randn(N, F, device="cuda", dtype=dtype)
    gb = X.numel() * X.element_size() / 1e9
    y = X[123].cpu().numpy()
    t = timeit(lambda: ld.dist(X, y, "euclidean"))
    print("%s %dx%d dist:            %7.2f ms  %.2f TB/s" % (str(dtype)[6:], N, F, t, gb / t))
    K = 20
    kc = KCenters(n_clusters=K, random_state=0)
    t = timeit(lambda: kc.fit([X]), 2)
    print("%s %dx%d KCenters K=%d fit: %7.2f ms  %.2f ms/pass  %.2f TB/s" % (str(dtype)[6:], N, F, K, t, t / K, gb * K / t))
    for K in (8, 100):
        Y = X[:K].cpu().numpy()
        t = timeit(lambda: ld.assign_nearest(X, Y, "euclidean"), 2)
        print("%s %dx%d assign K=%4d:     %7.2f ms  %.2f TB/s  %.2f T pair-elements/s" % (str(dtype)[6:], N, F, K, t, gb / t, N * F * K / t / 1e9))
    del X
