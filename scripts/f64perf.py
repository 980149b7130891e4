import ctypes as C, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmbuilder_amd import tICA, _lib
os.environ["MSMBUILDER_AMD_TICA_MODE"] = "f64"
for dtype, N in ((torch.float32, 2_000_000), (torch.float64, 1_000_000)):
    F, T = 512, 10000
    X = torch.randn(N, F, device="cuda", dtype=dtype)
    seqs = list(X.view(N // T, T, F).unbind(0))
    best = 1e9
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it in range(3):
            m = tICA(lag_time=100).fit(seqs)
            ms = C.c_float(); _lib.check(_lib.lib().msm_tica_last_kernel_ms(m._handle, C.byref(ms)))
            best = min(best, ms.value)
    print("f64 mode %s N=%d: kernel %.2f ms  %.1f TF alg (%.3f of 78.6)  %.1fM frames/s" % (
        str(dtype)[6:], N, best, 4.0 * F * F * N / best / 1e9, 4.0 * F * F * N / best / 1e9 / 78.6, N / best / 1e3))
    del X, seqs
