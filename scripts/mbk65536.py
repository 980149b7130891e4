"""MiniBatchKMeans(k=1000, batch_size=65536) on a [10M, 10] float32 projection (the bench's large-batch leg): wall time,
steps, and -- with PROFILE=1 -- the host-side cProfile of one fit."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import MiniBatchKMeans
warnings.simplefilter("ignore")
n = int(os.environ.get("N_ROWS", "10000000"))
g = torch.Generator(device="cuda").manual_seed(5)
X = (torch.randn(n, 10, generator=g, device="cuda") * torch.linspace(3, 0.3, 10, device="cuda")).float().contiguous()
B = int(os.environ.get("BATCH", "65536"))
def fit():
    torch.cuda.synchronize(); t = time.perf_counter()
    mb = MiniBatchKMeans(n_clusters=1000, batch_size=B, random_state=0).fit([X])
    torch.cuda.synchronize(); t = time.perf_counter() - t
    return mb, t
mb, t = fit()
for _ in range(2):
    mb, t = fit()
    print("MBKM K=1000 batch %d on %d x 10: fit %.1f ms, %d steps, %.1f us/step, %.1fM rows/s through the steps" % (
        B, n, 1e3 * t, mb.n_steps_, 1e6 * t / mb.n_steps_, mb.n_steps_ * B / t / 1e6))
if os.environ.get("PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); fit(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
