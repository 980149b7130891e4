"""Profiling helper: dir-npy -> HBM streaming rate and tICA.fit fed from it (PCIe-inclusive numbers)."""
import os, sys, time, tempfile, shutil, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA
from msmbuilder_amd.dataset import dataset

d = tempfile.mkdtemp(dir="/tmp")
try:
    ds = dataset(os.path.join(d, "ds"), mode="w")
    rs = np.random.RandomState(0)
    n_files, T, F = 96, 10000, 512
    base = rs.randn(T, F).astype(np.float32)
    for i in range(n_files):
        ds[i] = base + np.float32(i)
    ds = dataset(os.path.join(d, "ds"))
    gb = n_files * T * F * 4 / 1e9
    for prefetch, buf, rd in ((2, 64 << 20, 1), (4, 8 << 20, 4), (4, 8 << 20, 8), (6, 4 << 20, 16)):
        for rep in range(2):
            torch.cuda.synchronize(); t = time.perf_counter()
            n = 0
            for x in ds.device_sequences(prefetch=prefetch, buffer_bytes=buf, readers=rd):
                n += x.shape[0]
            torch.cuda.synchronize(); dt = time.perf_counter() - t
        print("stream %d files (%.2f GB) prefetch=%d buf=%dMiB readers=%d: %.3f s  %.2f GB/s  %.2fM frames/s" % (n_files, gb, prefetch, buf >> 20, rd, dt, gb / dt, n / dt / 1e6))
    warnings.simplefilter("ignore")
    for name, src in (("device stream", lambda: ds.device_sequences()), ("np.load + pageable staging", lambda: ds)):
        for rep in range(2):
            torch.cuda.synchronize(); t = time.perf_counter()
            m = tICA(n_components=10, lag_time=100).fit(src())
            m.eigenvalues_
            torch.cuda.synchronize(); dt = time.perf_counter() - t
        print("tICA.fit from dir-npy via %-28s: %.3f s  %.2fM frames/s" % (name, dt, n_files * T / dt / 1e6))
finally:
    shutil.rmtree(d, ignore_errors=True)
