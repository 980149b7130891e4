import os, sys, time, warnings
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from msmbuilder_amd import KCenters
warnings.simplefilter("ignore")
g = torch.Generator(device="cuda").manual_seed(1)
for (n, m, K, dt) in ((3_000_000, 100, 1000, torch.float32), (1_500_000, 60, 600, torch.float64), (400_000, 250, 300, torch.float32)):
    hubs = torch.randn(40, m, generator=g, device="cuda") * 3
    X = (hubs[torch.randint(0, 40, (n,), generator=g, device="cuda")] + torch.randn(n, m, generator=g, device="cuda")).to(dt).contiguous()
    res = {}
    for sw in ("1", "0"):
        os.environ["MSM_KC_WBATCH"] = sw
        torch.cuda.synchronize(); t = time.perf_counter()
        kc = KCenters(n_clusters=K, random_state=3).fit([X])
        torch.cuda.synchronize(); t = time.perf_counter() - t
        res[sw] = (list(kc.cluster_ids_), kc.labels_[0].cpu().numpy(), kc.distances_[0].cpu().numpy(), t)
    same = res["1"][0] == res["0"][0] and np.array_equal(res["1"][1], res["0"][1]) and np.array_equal(res["1"][2], res["0"][2])
    print("n=%d m=%d K=%d %s: batched %.1f ms, one centre per pass %.1f ms, identical: %s" % (n, m, K, str(dt)[6:], 1e3 * res["1"][3], 1e3 * res["0"][3], same), flush=True)
    del X
