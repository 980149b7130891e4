"""scripts/mbkprof.py -- where a MiniBatchKMeans(k=1000) fit on a [10M, 10] fp32 projection spends its time (cProfile of the
host side; the GPU side is in profiles/r05_pmc_all.txt).  Arguments: batch size (default 65536)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import MiniBatchKMeans
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
g = torch.Generator(device="cuda").manual_seed(3)
n = 10_000_000
cen = torch.randn(1000, 10, generator=g, device="cuda") * 3
Y = (cen[torch.randint(0, 1000, (n,), generator=g, device="cuda")] + torch.randn(n, 10, generator=g, device="cuda")).float().contiguous()
kw = dict(n_clusters=1000, random_state=0) if bs == 0 else dict(n_clusters=1000, random_state=0, batch_size=bs)
MiniBatchKMeans(**kw).fit([Y])
torch.cuda.synchronize()
for rep in range(2):
    t = time.perf_counter()
    m = MiniBatchKMeans(**kw).fit([Y])
    torch.cuda.synchronize()
    print("fit %.1f ms, %d steps" % (1e3 * (time.perf_counter() - t), m.n_steps_))
pr = cProfile.Profile()
pr.enable()
m = MiniBatchKMeans(**kw).fit([Y])
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
