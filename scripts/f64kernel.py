"""Profiling helper: the fp64 MFMA accumulation kernel on 2M x 512 float32 input (mode f64), HIP-event kernel time."""
import ctypes as C, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MSMBUILDER_AMD_TICA_MODE"] = "f64"
import torch
from msmbuilder_amd import tICA, _lib
warnings.simplefilter("ignore")
X = torch.randn(2_000_000, 512, device="cuda") + 1.0
seqs = list(X.view(-1, 10000, 512).unbind(0))
best = 1e9
for _ in range(4):
    m = tICA(lag_time=100).fit(seqs)
    ms = C.c_float(); _lib.check(_lib.lib().msm_tica_last_kernel_ms(m._handle, C.byref(ms))); best = min(best, ms.value)
print("tica_mfma_f64_kernel 2M x 512: %.2f ms" % best)
