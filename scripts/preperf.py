"""Profiling helper: column scan + scaling throughput (HBM-bound) on the bench shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd.preprocessing import StandardScaler

def timeit(fn, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) * 1e3)
    return best

for dtype, N, F in ((torch.float32, 10_000_000, 512), (torch.float64, 4_000_000, 128)):
    X = torch.randn(N, F, device="cuda", dtype=dtype) * 3 + 1
    seqs = list(X.view(N // 10000, 10000, F).unbind(0))
    gb = X.numel() * X.element_size() / 1e9
    sc = StandardScaler()
    t = timeit(lambda: sc.fit(seqs))
    print("%s %dx%d StandardScaler.fit:       %7.2f ms  %.2f TB/s" % (str(dtype)[6:], N, F, t, gb / t))
    t = timeit(lambda: sc.partial_transform(X))
    print("%s %dx%d StandardScaler.transform: %7.2f ms  %.2f TB/s (read+write)" % (str(dtype)[6:], N, F, t, 2 * gb / t))
    del X, seqs

from msmbuilder_amd.preprocessing import RobustScaler
X = torch.randn(10_000_000, 512, device="cuda") * 3 + 1
seqs = list(X.view(1000, 10000, 512).unbind(0))
rb = RobustScaler()
t = timeit(lambda: rb.fit(seqs), 2)
print("float32 10000000x512 RobustScaler.fit (count pass + 3 radix-select passes): %7.2f ms" % t)
Z = torch.randn(10_000_000, 16, device="cuda").cumsum(0) * 0.01
X = Z @ torch.randn(16, 512, device="cuda") + 0.3 * torch.randn(10_000_000, 512, device="cuda")
seqs = list(X.view(1000, 10000, 512).unbind(0))
t = timeit(lambda: rb.fit(seqs), 2)
print("float32 10000000x512 RobustScaler.fit on slowly varying features:          %7.2f ms" % t)
