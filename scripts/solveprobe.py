"""scripts/solveprobe.py -- which route the hybrid solve takes, and how long, when n_components reaches into the flat part of the
spectrum (fewer slow modes than components)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from msmbuilder_amd import tICA
warnings.simplefilter("ignore")
for F in (128, 512, 1024):
    for n_slow in (16, 6, 2):
        X = bench.synth(torch, 100, 10000, F, 7, "cuda", n_slow=n_slow)
        seqs = list(X.view(100, 10000, F).unbind(0))
        for k in (10,):
            ts = []
            for _ in range(6):
                m = tICA(n_components=k, lag_time=100).fit(seqs)
                torch.cuda.synchronize(); t = time.perf_counter()
                e = m.eigenvalues_
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
            print("F=%4d slow modes=%2d k=%d: solve %.2f ms  route %s  eigenvalues %s" % (F, n_slow, k, 1e3 * min(ts), getattr(m, "_solve_route", None), np.array2string(e, precision=4)))
        del X, seqs
