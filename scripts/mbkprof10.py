"""Profiling helper: MiniBatchKMeans(k=1000) on a [2M, 10] fp32 projection (the bench's MBKM leg shape) and on [1.25M, 512]."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import MiniBatchKMeans
warnings.simplefilter("ignore")
for n, F in ((2_000_000, 10), (1_250_000, 512)):
    X = torch.randn(n, F, device="cuda")
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        mb = MiniBatchKMeans(n_clusters=1000, random_state=0).fit([X])
        torch.cuda.synchronize(); t = time.perf_counter() - t
    print("MBKM K=1000 on %d x %d: %.1f ms, %d steps, %.1f us/step" % (n, F, 1e3 * t, mb.n_steps_, 1e6 * t / mb.n_steps_))
