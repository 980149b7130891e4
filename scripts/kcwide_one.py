"""one shape of scripts/kcwide.py for rocprofv3 --kernel-trace --stats: python scripts/kcwide_one.py n m K f32|f64"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmbuilder_amd import KCenters
warnings.simplefilter("ignore")
n, m, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dt = torch.float32 if sys.argv[4] == "f32" else torch.float64
g = torch.Generator(device="cuda").manual_seed(5)
cen = torch.randn(50, m, generator=g, device="cuda") * 3
X = (cen[torch.randint(0, 50, (n,), generator=g, device="cuda")] + torch.randn(n, m, generator=g, device="cuda")).to(dt).contiguous()
if n == 280_000 and m == 171:   # the bench leg's data (config3_stress): one blob of contact-like distances
    gC = torch.Generator(device="cuda").manual_seed(171)
    X = (torch.linspace(0.4, 2.5, 171, device="cuda") + 0.2 * torch.randn(n, 171, generator=gC, device="cuda")).abs().float().contiguous()
for _ in range(3):
    KCenters(n_clusters=K, random_state=0).fit([X])
torch.cuda.synchronize()
