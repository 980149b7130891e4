"""The bf16 image path alone (1M x 2048, lag 100, modes bf16 and bf16x2) for rocprofv3 kernel stats / PMC passes."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmbuilder_amd import tICA
warnings.simplefilter("ignore")
T, F, n_seq = 10000, 2048, 100
X = torch.randn(n_seq * T, F, device="cuda") + 2.0
seqs = list(X.view(n_seq, T, F).unbind(0))
for mode in (sys.argv[1:] or ["bf16", "bf16x2"]):
    os.environ["MSMBUILDER_AMD_TICA_MODE"] = mode
    for _ in range(2):
        m = tICA(n_components=5, lag_time=100).fit(seqs)
    print(mode, m.eigenvalues_[:2])
