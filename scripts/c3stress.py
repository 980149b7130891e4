"""SURVEY 8(d)'s C3 stress variant alone: KCenters(200).fit + predict on 280,000 x 171 float32 (28 x 10,000), no tICA."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmbuilder_amd import KCenters
warnings.simplefilter("ignore")
g = torch.Generator(device="cuda").manual_seed(171)
X = (torch.linspace(0.4, 2.5, 171, device="cuda") + 0.2 * torch.randn(280_000, 171, generator=g, device="cuda")).abs().float().contiguous()
seqs = list(X.view(28, 10_000, 171).unbind(0))
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    kc = KCenters(n_clusters=200, random_state=0).fit(seqs)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    lab = kc.predict(seqs)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("KCenters(200) on 280,000 x 171 fp32: fit %.2f ms (%.2f TB/s if every pass read X), predict %.2f ms (%.2f T pair-elements/s)" % (
        1e3 * (t1 - t), 200 * X.numel() * 4 / (t1 - t) / 1e12, 1e3 * (t2 - t1), 280_000 * 200 * 171 / (t2 - t1) / 1e12))
