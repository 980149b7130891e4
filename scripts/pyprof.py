import cProfile, pstats, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmbuilder_amd import tICA
X = torch.randn(1000 * 10000, 512, device="cuda")
seqs = list(X.view(1000, 10000, 512).unbind(0))
warnings.simplefilter("ignore")
tICA(lag_time=100).fit(seqs)
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    m = tICA(lag_time=100).fit(seqs)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
