"""Profiling helper: kernel-level throughput across the BASELINE configs' shapes."""
import ctypes as C, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA, KCenters, MiniBatchKMeans, _lib
from msmbuilder_amd.cluster.minibatchkmeans import label_inertia

def tica_case(N, F, T, lag, mode, dtype=torch.float32):
    os.environ["MSMBUILDER_AMD_TICA_MODE"] = mode
    n_seq = max(1, N // T)
    X = torch.randn(n_seq * T, F, device="cuda", dtype=dtype)
    seqs = list(X.view(n_seq, T, F).unbind(0))
    best = 1e9
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it in range(3):
            m = tICA(lag_time=lag).fit(seqs)
            ms = C.c_float(); _lib.check(_lib.lib().msm_tica_last_kernel_ms(m._handle, C.byref(ms)))
            best = min(best, ms.value)
    fl = 4.0 * F * F * n_seq * T
    print("tICA %-3s %s N=%9d F=%4d T=%6d lag=%3d : kernel %8.2f ms  %7.1f TF alg  %8.1fM frames/s" % (
        mode, str(dtype)[6:], n_seq * T, F, T, lag, best, fl / best / 1e9, n_seq * T / best / 1e3), flush=True)
    del X, seqs

def timeit(fn, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) * 1e3)
    return best

tica_case(1_000_000, 128, 1_000_000, 100, "f32")      # config 2
tica_case(1_000_000, 128, 10_000, 100, "f32")
tica_case(280_000, 171, 10_000, 1, "f32")             # config 3 featurisation shape
tica_case(10_000_000, 512, 10_000, 100, "f32")        # config 4 shard
tica_case(2_000_000, 2048, 10_000, 100, "f32")        # config 5 width
tica_case(2_000_000, 512, 10_000, 100, "f64")
tica_case(1_000_000, 512, 10_000, 100, "f64", torch.float64)
tica_case(100_000, 512, 10_000, 100, "f32")           # one small batch
tica_case(10_000, 512, 10_000, 100, "f32")            # one trajectory per call

# config 3: KCenters on 280k x 10
Y = torch.randn(280_000, 10, device="cuda", dtype=torch.float64)
kc = KCenters(n_clusters=200, random_state=0)
print("KCenters fit 280k x 10 f64 K=200: %.2f ms" % timeit(lambda: kc.fit([Y])))
print("KCenters predict 280k x 10:       %.2f ms" % timeit(lambda: kc.predict([Y])))
Yf = Y.float()
print("KCenters fit 280k x 10 f32 K=200: %.2f ms" % timeit(lambda: KCenters(n_clusters=200, random_state=0).fit([Yf])))
# config 4: k-means labelling K=1000 on 10M x 512 (2M here) and a minibatch fit
X = torch.randn(2_000_000, 512, device="cuda")
Cn = X[:1000].cpu().numpy().copy()
t = timeit(lambda: label_inertia(X, Cn), 2)
print("kmeans label 2M x 512, K=1000: %.2f ms  -> %.1fM frames/s, %.1f TF (2NKF)" % (t, 2e6 / t / 1e3, 2 * 2e6 * 1000 * 512 / t / 1e9))
t0 = time.perf_counter()
mb = MiniBatchKMeans(n_clusters=1000, init=Cn, n_init=1, batch_size=1024, max_iter=1, random_state=0, compute_labels=False, max_no_improvement=None)
mb.max_iter = 1
import msmbuilder_amd.cluster.minibatchkmeans as M
Xs = X[:200_000]
mb.fit([Xs]); torch.cuda.synchronize()
print("MiniBatchKMeans K=1000 batch=1024 on 200k x 512: %d steps in %.1f ms -> %.2f ms/step" % (mb.n_steps_, (time.perf_counter() - t0) * 1e3, (time.perf_counter() - t0) * 1e3 / mb.n_steps_))
# exact assign in feature space (stress): 200k x 171, K=200
Z = torch.randn(200_000, 171, device="cuda")
kc2 = KCenters(n_clusters=200, random_state=0)
print("KCenters fit 200k x 171 f32 K=200: %.2f ms ; predict %.2f ms" % (timeit(lambda: kc2.fit([Z]), 1), timeit(lambda: kc2.predict([Z]), 1)))
