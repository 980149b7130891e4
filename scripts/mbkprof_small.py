"""Profiling helper: MiniBatchKMeans(k=1000) on a [2M, 10] fp32 projection only (the bench's MBKM leg shape), with phase times."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import MiniBatchKMeans
warnings.simplefilter("ignore")
X = torch.randn(2_000_000, 10, device="cuda")
for _ in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    mb = MiniBatchKMeans(n_clusters=1000, random_state=0).fit([X])
    torch.cuda.synchronize(); t = time.perf_counter() - t
    print("MBKM K=1000 on 2M x 10: %.1f ms, %d steps, %.1f us/step" % (1e3 * t, mb.n_steps_, 1e6 * t / mb.n_steps_))
