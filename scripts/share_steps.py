"""One rank's share of the bench at N = 8 (125 trajectories), the bench's step repeated: per-step phase times and the MFMA
kernel's HIP-event time, to see what a step costs once the clocks have settled (and what precedes a slow kernel)."""
import ctypes as C, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from msmbuilder_amd import tICA, KCenters, _lib
warnings.simplefilter("ignore")
n_seq, T, F = 125, 10000, 512
X = bench.synth(torch, n_seq, T, F, 1234, torch.device("cuda"))
seqs = list(X.view(n_seq, T, F).unbind(0))
KCenters._force_sharded = True
def kms(m):
    ms = C.c_float(); _lib.check(_lib.lib().msm_tica_last_kernel_ms(m._handle, C.byref(ms))); return ms.value
mode = sys.argv[1] if len(sys.argv) > 1 else "step"
rows = []
for it in range(16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = tICA(n_components=10, lag_time=100).fit(seqs)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    k = kms(m)
    if mode == "step":
        ev = m.eigenvalues_; comps = m.components_
        torch.cuda.synchronize(); t2 = time.perf_counter()
        Y = m.transform([X])[0]
        torch.cuda.synchronize(); t3 = time.perf_counter()
        kc = KCenters(n_clusters=200, random_state=0).fit([Y])
        torch.cuda.synchronize(); t4 = time.perf_counter()
        lab = kc.predict([Y])[0]
        torch.cuda.synchronize(); t5 = time.perf_counter()
        rows.append((k, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t5 - t4)))
    elif mode == "sleep":
        time.sleep(0.005)
        rows.append((k, 1e3 * (t1 - t0)))
    else:
        rows.append((k, 1e3 * (t1 - t0)))
print(mode, "kernel ms | fit | solve | transform | kcenters_fit | predict")
for r in rows:
    print("  " + "  ".join("%7.3f" % v for v in r))
a = np.array(rows[4:])
print("median after 4 warm-up steps:", "  ".join("%7.3f" % v for v in np.median(a, 0)), "  step %.3f" % np.median(a[:, 1:].sum(1)))
