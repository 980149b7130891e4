import ctypes as C, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA, _lib
import bench
warnings.simplefilter("ignore")
# accuracy on a small asymmetric problem vs fp64
rs = np.random.RandomState(0)
seqs = [(rs.randn(n, 200).cumsum(0) * 0.02 + rs.randn(n, 200) + np.linspace(-2, 2, 200)).astype(np.float32) for n in (3000, 1777, 40, 5000)]
ref = None
for mode in ("f64", "f32", "bf16x2", "bf16"):
    os.environ["MSMBUILDER_AMD_TICA_MODE"] = mode
    m = tICA(n_components=5, lag_time=7).fit(seqs); m._pull()
    if ref is None: ref = (m._outer_0_to_T_lagged.copy(), m._outer_gram_sum.copy(), m.eigenvalues_.copy())
    sc = np.abs(ref[1]).max()
    print("%-7s max|dC|/scale %.2e  max|dG|/scale %.2e  eig rel err %.2e" % (mode, np.abs(m._outer_0_to_T_lagged - ref[0]).max() / sc,
          np.abs(m._outer_gram_sum - ref[1]).max() / sc, np.abs(m.eigenvalues_ / ref[2] - 1).max()))
# speed on the bench workload
X = bench.synth(torch, 1000, 10000, 512, 1234, torch.device("cuda"))
sq = list(X.view(1000, 10000, 512).unbind(0))
evs = {}
for mode in ("f32", "bf16x2", "bf16"):
    os.environ["MSMBUILDER_AMD_TICA_MODE"] = mode
    best = 1e9
    for it in range(3):
        m = tICA(n_components=10, lag_time=100).fit(sq)
        ms = C.c_float(); _lib.check(_lib.lib().msm_tica_last_kernel_ms(m._handle, C.byref(ms))); best = min(best, ms.value)
    evs[mode] = m.eigenvalues_
    print("%-7s kernel %.2f ms  %.1f TF alg  %.1fM frames/s   eig rel diff vs f32 %.2e" % (mode, best, 4 * 512 * 512 * 1e7 / best / 1e9, 1e7 / best / 1e3,
          np.abs(evs[mode] / evs["f32"] - 1).max()))
