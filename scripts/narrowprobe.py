"""scripts/narrowprobe.py -- tICA.fit / transform on NARROW feature sets (4 .. 96 features), 8M frames: bytes per second against HBM."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from msmbuilder_amd import tICA, _lib
warnings.simplefilter("ignore")
g = torch.Generator(device="cuda").manual_seed(1)
N, T = 8_000_000, 10000
for F in (4, 8, 16, 32, 64, 96, 128):
    for dt in (torch.float32, torch.float64):
        X = torch.randn(N, F, generator=g, device="cuda").to(dt)
        seqs = list(X.view(N // T, T, F).unbind(0))
        for _ in range(2):
            m = tICA(n_components=min(4, F), lag_time=10).fit(seqs)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            m = tICA(n_components=min(4, F), lag_time=10).fit(seqs)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        km = bench.kernel_ms_of(m, _lib)
        _ = m.eigenvalues_
        tt = []
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            Y = m.transform(seqs)
            torch.cuda.synchronize(); tt.append(time.perf_counter() - t)
        gb = N * F * X.element_size() / 1e9
        print("F=%3d %-7s fit %7.2f ms (kernel %6.2f ms = %5.2f TB/s of rows)   transform %6.2f ms = %5.2f TB/s" % (F, str(dt)[6:], 1e3 * min(ts), km, gb / km, 1e3 * min(tt), gb / (1e3 * min(tt))))
        del X, seqs, Y
