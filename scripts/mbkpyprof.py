"""Profiling helper: cProfile of MiniBatchKMeans(k=1000).fit on a [2M, 10] fp32 projection (host-side costs of the step loop)."""
import cProfile, pstats, os, sys, warnings, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmbuilder_amd import MiniBatchKMeans
warnings.simplefilter("ignore")
X = torch.randn(2_000_000, 10, device="cuda")
MiniBatchKMeans(n_clusters=1000, random_state=0).fit([X])
pr = cProfile.Profile(); pr.enable()
mb = MiniBatchKMeans(n_clusters=1000, random_state=0).fit([X])
torch.cuda.synchronize()
pr.disable()
print("steps", mb.n_steps_)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
