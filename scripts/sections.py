"""Profiling helper (needs a -DMSM_TICA_PROFILE build): per-section shader cycles of the fp32 MFMA kernel."""
import ctypes as C, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmbuilder_amd import tICA, _lib
F, T, n_seq, lag = 512, 10000, 1000, 100
X = torch.randn(n_seq * T, F, device="cuda")
seqs = list(X.view(n_seq, T, F).unbind(0))
warnings.simplefilter("ignore")
for it in range(2):
    m = tICA(lag_time=lag).fit(seqs)
ms = C.c_float(); _lib.check(_lib.lib().msm_tica_last_kernel_ms(m._handle, C.byref(ms)))
out = (C.c_int64 * 64)()
_lib.check(_lib.lib().msm_tica_debug_profile(m._handle, out))
print("kernel %.2f ms; wave-0 clocks %d" % (ms.value, out[1] - out[0]))
names = (["chunk prologue", "first fragments", "MFMA stream", "barrier", "final", "slab merge + chunk switch"] if os.environ.get("MSM_TICA_SYM", "1") != "0"
         else ["chunk prologue", "step head", "MFMA loop", "step tail", "final merge", "inter-chunk merge"])
for slot in range(5):
    v = [out[8 + 8 * slot + i] for i in range(6)]
    tot = float(sum(v)) or 1.0
    print("slot %d: " % slot + "  ".join("%s %.1f%%" % (n, 100 * x / tot) for n, x in zip(names, v)) + "   total %.3g cycles" % tot)
