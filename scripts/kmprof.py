"""k-means labelling shapes for rocprofv3 (config 4: K=1000, F=512)."""
import sys, time, numpy as np, torch
from msmbuilder_amd.cluster.minibatchkmeans import label_inertia
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
torch.manual_seed(0)
X = torch.randn(N, 512, device="cuda")
Cn = torch.randn(1000, 512).numpy()
for _ in range(3):
    t = time.perf_counter(); label_inertia(X, Cn); torch.cuda.synchronize(); print("%.2f ms" % (1e3 * (time.perf_counter() - t)))
