"""Profiling helper: time the tICA MFMA kernel alone (HIP events) under MSM_TICA_ABLATE masks."""
import ctypes as C, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmbuilder_amd import tICA, _lib

F = int(os.environ.get("F", 512)); T = 10000; n_seq = int(os.environ.get("NSEQ", 1000)); lag = 100
X = torch.randn(n_seq * T, F, device="cuda")
seqs = list(X.view(n_seq, T, F).unbind(0))
for mask in [int(a) for a in sys.argv[1:]] or [0]:
    os.environ["MSM_TICA_ABLATE"] = str(mask)
    ts = []
    for it in range(3):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = tICA(lag_time=lag).fit(seqs)
        ms = C.c_float()
        _lib.check(_lib.lib().msm_tica_last_kernel_ms(m._handle, C.byref(ms)))
        ts.append(ms.value)
        clk = (C.c_int64 * 4)()
        _lib.check(_lib.lib().msm_tica_debug_clocks(m._handle, clk))
    t = min(ts)
    ghz = (clk[1] - clk[0]) / ((clk[3] - clk[2]) * 10.0)  # shader cycles per ns (wall clock = 100 MHz)
    print("ablate=%d  clk %.3f GHz  kernel %.2f ms  %.1f TF algorithmic (%.3f of 157.3)  %.1fM frames/s" % (
        mask, ghz, t, 4 * F * F * n_seq * T / t / 1e9, 4 * F * F * n_seq * T / t / 1e9 / 157.3, n_seq * T / t / 1e3))
