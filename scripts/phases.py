"""Profiling helper: fine-grained wall times of the bench pipeline phases (device-synchronised)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA, KCenters, _lib
import bench

F, T, n_seq = 512, 10000, 1000
X = bench.synth(torch, n_seq, T, F, 1234, torch.device("cuda"))
seqs = list(X.view(n_seq, T, F).unbind(0))
def tm(label, fn, acc):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    acc.setdefault(label, []).append((time.perf_counter() - t) * 1e3); return r
acc = {}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for it in range(4):
        tica = tICA(n_components=10, lag_time=100)
        tm("tica.fit", lambda: tica.fit(seqs), acc)
        tm("pull(export)", lambda: tica._pull(), acc)
        tm("solve", lambda: tica.eigenvalues_, acc)
        Y = tm("transform", lambda: tica.transform([X])[0], acc)
        kc = KCenters(n_clusters=200, random_state=0)
        Yc = tm("kc._concat", lambda: kc._concat([Y]), acc)
        from msmbuilder_amd.cluster.kcenters import _KCenters
        tm("kc.fit(single array)", lambda: _KCenters.fit(kc, Yc), acc)
        tm("kc.predict", lambda: _KCenters.predict(kc, Y), acc)
for k, v in acc.items():
    print("%-24s %8.2f ms (min %.2f)" % (k, np.mean(v[1:]), min(v)))

# raw C-ABI call timing of the k-centers driver
import ctypes as C
L = _lib.lib()
n = Yc.shape[0]
ids = np.zeros(200, dtype=np.int64)
lab = torch.empty(n, dtype=torch.int64, device="cuda"); dist = torch.empty(n, dtype=torch.float64, device="cuda")
inertia = C.c_double()
for K in (200, 50):
    ts = []
    for it in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        _lib.check(L.msm_kcenters_fit_f64(C.c_void_p(Yc.data_ptr()), n, 10, K, b"euclidean", 0, ids.ctypes.data,
                                          C.c_void_p(lab.data_ptr()), C.c_void_p(dist.data_ptr()), C.byref(inertia), 1))
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print("raw msm_kcenters_fit_f64 K=%d: %s ms" % (K, ["%.1f" % x for x in ts]))

from msmbuilder_amd._lib import Arr, empty_like_placement
for it in range(3):
    t0 = time.perf_counter(); ax = Arr(Yc); torch.cuda.synchronize(); t1 = time.perf_counter()
    labels = empty_like_placement(ax, (n,), np.int64); distances = empty_like_placement(ax, (n,), np.float64)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    al, ad = Arr(labels, np.int64), Arr(distances, np.float64); t3 = time.perf_counter()
    cc = ax.keep[[int(i) for i in ids]]; torch.cuda.synchronize(); t4 = time.perf_counter()
    from sklearn.utils import check_random_state
    s = check_random_state(0).randint(0, n); t5 = time.perf_counter()
    print("Arr %.2f  empty %.2f  Arr2 %.2f  index %.2f  rng %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t5-t4)*1e3))
    del labels, distances

print("fresh outputs each call:")
for it in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    lab2 = torch.empty(n, dtype=torch.int64, device="cuda"); dist2 = torch.empty(n, dtype=torch.float64, device="cuda")
    _lib.check(L.msm_kcenters_fit_f64(C.c_void_p(Yc.data_ptr()), n, 10, 200, b"euclidean", 0, ids.ctypes.data,
                                      C.c_void_p(lab2.data_ptr()), C.c_void_p(dist2.data_ptr()), C.byref(inertia), 1))
    torch.cuda.synchronize(); print("  %.1f ms  ptr %x" % ((time.perf_counter() - t) * 1e3, lab2.data_ptr()))
    del lab2, dist2
print("via _KCenters.fit on one object:")
kc = KCenters(n_clusters=200, random_state=0)
for it in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    _KCenters.fit(kc, Yc)
    torch.cuda.synchronize(); print("  %.1f ms  ptr %x" % ((time.perf_counter() - t) * 1e3, kc.labels_.data_ptr()))

print("idle-gap experiment (sleep before fit):")
for gap in (0.0, 0.005, 0.02, 0.05, 0.2):
    ts = []
    for it in range(3):
        torch.cuda.synchronize(); time.sleep(gap); t = time.perf_counter()
        _KCenters.fit(kc, Yc)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print("  gap %.0f ms -> fit %s ms" % (gap * 1e3, ["%.1f" % x for x in ts]))

print("pipeline loop, raw C call for k-centers:")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for it in range(4):
        tica = tICA(n_components=10, lag_time=100)
        tica.fit(seqs); ev = tica.eigenvalues_
        Y = tica.transform([X])[0]
        Yc2 = torch.cat([Y], dim=0).contiguous()
        lab2 = torch.empty(n, dtype=torch.int64, device="cuda"); dist2 = torch.empty(n, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize(); t = time.perf_counter()
        _lib.check(L.msm_kcenters_fit_f64(C.c_void_p(Yc2.data_ptr()), n, 10, 200, b"euclidean", 0, ids.ctypes.data,
                                          C.c_void_p(lab2.data_ptr()), C.c_void_p(dist2.data_ptr()), C.byref(inertia), 1))
        torch.cuda.synchronize(); t1 = (time.perf_counter() - t) * 1e3
        torch.cuda.synchronize(); t = time.perf_counter()
        _lib.check(L.msm_kcenters_fit_f64(C.c_void_p(Yc2.data_ptr()), n, 10, 200, b"euclidean", 0, ids.ctypes.data,
                                          C.c_void_p(lab2.data_ptr()), C.c_void_p(dist2.data_ptr()), C.byref(inertia), 1))
        torch.cuda.synchronize(); t2 = (time.perf_counter() - t) * 1e3
        print("  first %.1f ms, repeat %.1f ms, Yc ptr %x, ids[:4] %s inertia %.6f" % (t1, t2, Yc2.data_ptr(), ids[:4], inertia.value))
