"""scripts/clusterprobe.py -- clusterers on shapes the bench does not have (wall times; anomalies show as outliers)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import KCenters, MiniBatchKMeans
warnings.simplefilter("ignore")
g = torch.Generator(device="cuda").manual_seed(5)
def data(n, m, kc, dtype):
    cen = torch.randn(kc, m, generator=g, device="cuda") * 3
    return (cen[torch.randint(0, kc, (n,), generator=g, device="cuda")] + torch.randn(n, m, generator=g, device="cuda")).to(dtype).contiguous()
def timeit(tag, f, reps=3):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print("%-70s %9.2f ms" % (tag, 1e3 * min(ts)))
    return r
for (n, m, K, dt) in ((10_000_000, 10, 200, torch.float64), (10_000_000, 10, 1000, torch.float64), (2_000_000, 10, 200, torch.float32),
                      (1_000_000, 64, 200, torch.float32), (1_000_000, 171, 500, torch.float32), (500_000, 512, 200, torch.float32),
                      (2_000_000, 3, 100, torch.float64), (2_000_000, 16, 200, torch.float64), (2_000_000, 17, 200, torch.float64)):
    X = data(n, m, 50, dt)
    for metric in ("euclidean",) + (("cityblock",) if m == 64 else ()):
        kc = timeit("KCenters(%d, %s).fit %d x %d %s" % (K, metric, n, m, str(dt)[6:]), lambda: KCenters(n_clusters=K, metric=metric, random_state=0).fit([X]))
        timeit("   predict", lambda: kc.predict([X]))
    del X
for (n, m, K, bs) in ((1_000_000, 64, 100, 1024), (1_000_000, 512, 1000, 1024), (1_000_000, 512, 1000, 8192), (4_000_000, 10, 100, 1024), (1_000_000, 30, 2000, 4096)):
    X = data(n, m, 80, torch.float32)
    mb = timeit("MiniBatchKMeans(%d, batch %d).fit %d x %d f32" % (K, bs, n, m), lambda: MiniBatchKMeans(n_clusters=K, batch_size=bs, random_state=0).fit([X]), reps=2)
    print("      steps %d" % mb.n_steps_)
    del X
