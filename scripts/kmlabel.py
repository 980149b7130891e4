"""Profiling helper: k-means labelling 2M x 512 x K = 1000 (kmeans_label_v4_kernel), wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd.cluster.minibatchkmeans import label_inertia
X = torch.randn(2_000_000, 512, device="cuda"); Ck = torch.randn(1000, 512).numpy()
ts=[]
for _ in range(6):
    torch.cuda.synchronize(); t=time.perf_counter(); label_inertia(X, Ck); torch.cuda.synchronize(); ts.append(time.perf_counter()-t)
print("kmeans label 2M x 512 K=1000: %.2f ms" % (1e3*min(ts[1:])))
