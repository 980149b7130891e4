"""PCIe-inclusive rate: tICA.fit on HOST (numpy) trajectories, 2M x 512 fp32 = 4.1 GB."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from msmbuilder_amd import tICA
F, T, n_seq = 512, 10000, 200
rs = np.random.RandomState(0)
X = np.random.default_rng(0).standard_normal((n_seq * T, F), dtype=np.float32)
seqs = [X[i * T:(i + 1) * T] for i in range(n_seq)]
warnings.simplefilter("ignore")
for it in range(3):
    t = time.perf_counter(); m = tICA(n_components=10, lag_time=100).fit(seqs); m.n_observations_; dt = time.perf_counter() - t
    print("host fit %.1f ms  %.2fM frames/s  %.1f GB/s" % (1e3 * dt, n_seq * T / dt / 1e6, X.nbytes / dt / 1e9))
if os.environ.get("PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); tICA(n_components=10, lag_time=100).fit(seqs).n_observations_; pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(8)
from msmbuilder_amd.preprocessing import StandardScaler
sc = StandardScaler().fit(seqs)
for it in range(2):
    t = time.perf_counter(); out = sc.transform(seqs); dt = time.perf_counter() - t
    print("host StandardScaler.transform %.1f ms  %.1f GB/s in + %.1f GB/s out" % (1e3 * dt, X.nbytes / dt / 1e9, X.nbytes / dt / 1e9))
for it in range(2):
    t = time.perf_counter(); Y = m.transform(seqs); dt = time.perf_counter() - t
    print("host tICA.transform %.1f ms  %.2fM frames/s" % (1e3 * dt, n_seq * T / dt / 1e6))
