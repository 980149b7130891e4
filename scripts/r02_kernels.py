"""Driver for rocprofv3 passes over the secondary kernels VERDICT r1 asked evidence for: the fp64 and bf16 tICA kernels,
exact assign_nearest (short rows: 2 rows / lane; wide rows), one k-centers pass, k-means labelling (K = 1000)."""
import ctypes as C, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA, KCenters, libdistance, _lib
from msmbuilder_amd.cluster.minibatchkmeans import label_inertia
warnings.simplefilter("ignore")
def kms(m):
    ms = C.c_float(); _lib.check(_lib.lib().msm_tica_last_kernel_ms(m._handle, C.byref(ms))); return ms.value
T = 10000
# tICA fp64 kernel, 2M x 512 (float32 input, the reference's arithmetic)
X = torch.randn(2_000_000, 512, device="cuda") + 1.0
seqs = list(X.view(-1, T, 512).unbind(0))
os.environ["MSMBUILDER_AMD_TICA_MODE"] = "f64"
for _ in range(2): m = tICA(lag_time=100).fit(seqs)
print("tica_mfma_f64_kernel 2M x 512: %.2f ms" % kms(m)); del m
os.environ["MSMBUILDER_AMD_TICA_MODE"] = "f32"
del X, seqs
# tICA at F = 2048 (configs[4] width): fp32 sum/difference, bf16x2, bf16
X = torch.randn(1_000_000, 2048, device="cuda") + 1.0
seqs = list(X.view(-1, T, 2048).unbind(0))
for mode in ("f32", "bf16x2", "bf16"):
    os.environ["MSMBUILDER_AMD_TICA_MODE"] = mode
    for _ in range(2): m = tICA(lag_time=100).fit(seqs)
    print("F=2048 mode %s 1M frames: %.2f ms" % (mode, kms(m))); del m
os.environ["MSMBUILDER_AMD_TICA_MODE"] = "f32"
del X, seqs
# exact assign_nearest: KCenters.predict shape and wide rows; one k-centers fit of 20 passes on wide rows
def tm(f, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return 1e3 * best
Y = torch.randn(10_000_000, 10, device="cuda", dtype=torch.float64)
Cn = Y[:200].cpu().numpy()
print("assign_nearest 10M x 10 f64 K=200: %.2f ms" % tm(lambda: libdistance.assign_nearest(Y, Cn, "euclidean")))
print("KCenters(20).fit 10M x 10 f64: %.2f ms" % tm(lambda: KCenters(n_clusters=20, random_state=0).fit([Y])))
del Y
X = torch.randn(4_000_000, 512, device="cuda")
Cn = X[:100].cpu().numpy()
print("assign_nearest 4M x 512 f32 K=100: %.2f ms" % tm(lambda: libdistance.assign_nearest(X, Cn, "euclidean"), 2))
print("KCenters(8).fit 4M x 512 f32: %.2f ms" % tm(lambda: KCenters(n_clusters=8, random_state=0).fit([X]), 2))
# k-means labelling K = 1000, F = 512
Ck = torch.randn(1000, 512).numpy()
print("kmeans label 2M x 512 K=1000: %.2f ms" % tm(lambda: label_inertia(X[:2_000_000], Ck)))
