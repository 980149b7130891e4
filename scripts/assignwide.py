"""Profiling helper: one exact assign_nearest at 4M x 512 float32 x K = 100 (the wide kernel), twice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmbuilder_amd import libdistance
n, m, K = 4_000_000, 512, 100
X = torch.randn(n, m, device="cuda", dtype=torch.float32)
Y = X[torch.randperm(n, device="cuda")[:K]].cpu().numpy()
for _ in range(2):
    libdistance.assign_nearest(X, Y, "euclidean")
torch.cuda.synchronize()
