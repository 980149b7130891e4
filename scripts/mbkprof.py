"""cProfile of one MiniBatchKMeans.fit at config 4's per-GPU shape: where the host time goes."""
import cProfile, os, pstats, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmbuilder_amd.cluster import MiniBatchKMeans
N = 1_250_000
torch.manual_seed(0)
X = (torch.randn(N, 16, device="cuda") @ torch.randn(16, 512, device="cuda") + 0.5 * torch.randn(N, 512, device="cuda")).contiguous()
warnings.simplefilter("ignore")
MiniBatchKMeans(n_clusters=1000, random_state=0, compute_labels=False).fit([X])
pr = cProfile.Profile(); pr.enable()
m = MiniBatchKMeans(n_clusters=1000, random_state=0, compute_labels=False).fit([X])
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
