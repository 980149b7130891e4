"""scripts/c2solve.py -- BASELINE configs[1] (1M x 128, one trajectory, lag 100): fit + solve wall time per solve route."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = torch.Generator(device="cuda").manual_seed(99)
z = torch.randn(1_000_000, 8, generator=g, device="cuda").cumsum(0) * 0.01
X = (z @ torch.randn(8, F, generator=g, device="cuda") + torch.randn(1_000_000, F, generator=g, device="cuda")).float().contiguous()
warnings.simplefilter("ignore")
for route in ("auto", "0", "hybrid", "auto", "0", "hybrid"):
    os.environ["MSMBUILDER_AMD_DEVICE_SOLVE"] = route
    ts, tf = [], []
    for _ in range(12):
        torch.cuda.synchronize(); t = time.perf_counter()
        m = tICA(n_components=10, lag_time=100).fit([X])
        torch.cuda.synchronize(); t1 = time.perf_counter()
        e = m.eigenvalues_
        torch.cuda.synchronize(); t2 = time.perf_counter()
        tf.append(t1 - t); ts.append(t2 - t1)
    print("route %-6s  fit %.3f ms  solve %.3f ms  (min of 12)  top %.6f" % (route, 1e3 * min(tf), 1e3 * min(ts), e[0]))
