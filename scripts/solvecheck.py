"""Profiling helper: tICA._solve wall time per mode (host / hybrid / device) on a fitted 512-feature model."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA
F = int(sys.argv[1]) if len(sys.argv) > 1 else 512
X = torch.randn(200000, 16, device="cuda") @ torch.randn(16, F, device="cuda") + 0.5 * torch.randn(200000, F, device="cuda") + 3.0
X[1:] = 0.7 * X[:-1] + 0.3 * X[1:]
seqs = list(X.view(20, 10000, F).unbind(0))
res = {}
for mode in ("0", "hybrid", "1"):
    os.environ["MSMBUILDER_AMD_DEVICE_SOLVE"] = mode
    ts = []
    for it in range(6):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = tICA(n_components=10, lag_time=100).fit(seqs)
            torch.cuda.synchronize()
            t = time.perf_counter()
            ev = m.eigenvalues_; V = m.eigenvectors_; mu = m.means_
            ts.append(time.perf_counter() - t)
    res[mode] = (ev.copy(), V.copy(), mu.copy(), m.shrinkage_)
    print("F=%d solve mode %-6s: %.2f ms (min of 5)  ev[:3]=%s shrinkage=%.6e" % (F, mode, 1e3 * min(ts[1:]), ev[:3], m.shrinkage_))
for mode in ("hybrid", "1"):
    e0, V0, mu0, s0 = res["0"]; e1, V1, mu1, s1 = res[mode]
    sgn = np.sign((V0 * V1).sum(0))
    print("mode %s vs host: eig rel %.2e  vec abs %.2e  mu rel %.2e  shrink rel %.2e" % (
        mode, np.abs(e1 / e0 - 1).max(), np.abs(V1 * sgn - V0).max() / np.abs(V0).max(), np.abs(mu1 / mu0 - 1).max(), abs(s1 / s0 - 1)))
