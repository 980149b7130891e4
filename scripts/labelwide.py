"""BASELINE configs[3]'s wide clusterer, one rank's share: 1.25M x 512 fp32 rows labelled against K = 1000 centres
(kmeans_label_v4_kernel + kmeans_inertia_kernel), with and without the XCD-grouped centre split (MSM_LABEL_XCD)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd.cluster.minibatchkmeans import label_inertia
n, m, K = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000, 512, 1000
g = torch.Generator(device="cuda").manual_seed(2)
Cn = torch.randn(K, m, generator=g, device="cuda") * 2
X = Cn[torch.randint(0, K, (n,), generator=g, device="cuda")] + torch.randn(n, m, generator=g, device="cuda")
Ch = Cn.cpu().numpy()
out = {}
for mode in ("0", "1", "2", "4", "0", "1", "2", "4"):
    os.environ["MSM_LABEL_XCD"] = mode
    label_inertia(X, Ch)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        lab, inertia = label_inertia(X, Ch)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    ms = 1e3 * min(ts)
    flop = 2.0 * n * 1024 * m          # executed: 8 centre tiles of 128
    print("MSM_LABEL_XCD=%s: label + inertia %.2f ms  (%.1f TF executed incl. the inertia pass and the centre upload = %.3f of the fp32 MFMA peak)"
          % (mode, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3))
    out[mode] = (lab.clone(), inertia)
print("labels identical:", bool(torch.equal(out["0"][0], out["1"][0])), " inertia rel diff %.2e" % abs(out["0"][1] / out["1"][1] - 1))
