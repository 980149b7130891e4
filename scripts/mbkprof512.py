"""Profiling helper: MiniBatchKMeans(k=1000, batch 1024) on [1.25M, 512] fp32 -- the F = 512 small-batch step."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmbuilder_amd import MiniBatchKMeans
warnings.simplefilter("ignore")
X = torch.randn(1_250_000, 512, device="cuda")
for _ in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    mb = MiniBatchKMeans(n_clusters=1000, random_state=0).fit([X])
    torch.cuda.synchronize(); t = time.perf_counter() - t
    print("MBKM K=1000 on 1.25M x 512: %.1f ms, %d steps, %.1f us/step all in" % (1e3 * t, mb.n_steps_, 1e6 * t / mb.n_steps_))
