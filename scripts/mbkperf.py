"""MiniBatchKMeans.fit at BASELINE config 4's per-GPU shape (1.25M x 512 fp32, K = 1000): wall time, steps, ms/step."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd.cluster import MiniBatchKMeans
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
torch.manual_seed(0)
Z = torch.randn(N, 16, device="cuda")
X = (Z @ torch.randn(16, 512, device="cuda") + 0.5 * torch.randn(N, 512, device="cuda")).contiguous()
warnings.simplefilter("ignore")
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    m = MiniBatchKMeans(n_clusters=1000, random_state=0, compute_labels=False).fit([X])
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("fit %.1f ms, %d steps -> %.3f ms/step (incl. init); inertia/n %.4f" % (1e3 * dt, m.n_steps_, 1e3 * dt / m.n_steps_, m.inertia_ / N))
t = time.perf_counter(); m = MiniBatchKMeans(n_clusters=1000, random_state=0).fit([X]); torch.cuda.synchronize()
print("fit + labels_ %.1f ms" % (1e3 * (time.perf_counter() - t)))
