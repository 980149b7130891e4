"""scripts/widthprobe.py -- tICA.fit (fp32, device rows) over feature widths from 4 to 384: kernel time, executed MFMA rate and
bytes of rows per second, for the whole-matrix sum/difference kernel (tica_symw_dev.h) and, with MSM_TICA_SYMW=0 in the
environment, the 128-wide kernels of rounds 1-5."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from msmbuilder_amd import tICA, _lib
warnings.simplefilter("ignore")
g = torch.Generator(device="cuda").manual_seed(1)
T = 10000
print("# MSM_TICA_SYMW=%s" % os.environ.get("MSM_TICA_SYMW", "(default: on)"))
for F, N, lag in ((4, 8_000_000, 10), (8, 8_000_000, 10), (16, 8_000_000, 10), (32, 8_000_000, 10), (64, 8_000_000, 10), (96, 8_000_000, 10),
                  (128, 8_000_000, 10), (128, 1_000_000, 100), (171, 2_000_000, 100), (192, 2_000_000, 100), (256, 2_000_000, 100),
                  (300, 2_000_000, 100), (384, 2_000_000, 100)):
    X = torch.randn(N, F, generator=g, device="cuda")
    seqs = list(X.view(N // T, T, F).unbind(0))
    for _ in range(2):
        m = tICA(n_components=min(4, F), lag_time=lag).fit(seqs)
    ts, ks = [], []
    for _ in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        m = tICA(n_components=min(4, F), lag_time=lag).fit(seqs)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        ks.append(bench.kernel_ms_of(m, _lib))
    km = min(ks)
    gb = N * F * 4 / 1e9
    print("F=%3d N=%8d lag %3d  fit %7.3f ms  kernel %7.3f ms = %6.1fM frames/s = %5.2f TB/s of rows, algorithmic 4 F^2: %6.1f TF"
          % (F, N, lag, 1e3 * min(ts), km, N / km / 1e3, gb / km, 4.0 * F * F * N / km / 1e9))
    del X, seqs
