"""The device-tail solve alone, for rocprofv3 --kernel-trace --stats (per-kernel durations of one tICA._solve at F = 512)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA
warnings.simplefilter("ignore")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 512
g = torch.Generator(device="cuda"); g.manual_seed(0)
z = torch.cumsum(torch.randn(200000, 16, generator=g, device="cuda"), 0) * 0.01
X = (torch.tanh(z - z.mean(0)) @ torch.randn(16, F, generator=g, device="cuda") + 0.5 * torch.randn(200000, F, generator=g, device="cuda")).float()
m = tICA(n_components=10, lag_time=100).fit(list(X.view(20, 10000, F).unbind(0)))
for _ in range(20):
    m._is_dirty = True
    m.eigenvalues_
print(m.eigenvalues_[:3])
