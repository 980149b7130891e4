"""Profiling helper: time the tICA fp32 MFMA kernel alone (HIP events) at a few widths."""
import ctypes as C, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmbuilder_amd import tICA, _lib
T, lag = 10000, 100
WIDTHS = [(int(a), 1000 if int(a) <= 768 else 200) for a in sys.argv[1:]] or [(512, 1000), (128, 1000), (2048, 200), (500, 1000)]
for F, n_seq in WIDTHS:
    X = torch.randn(n_seq * T, F, device="cuda")
    seqs = list(X.view(n_seq, T, F).unbind(0))
    ts = []
    for it in range(4):
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
    print("F=%d clk %.3f GHz  kernel %.2f ms  %.1f TF algorithmic (%.3f of 157.3)  %.1fM frames/s" % (
        F, ghz, t, 4 * F * F * n_seq * T / t / 1e9, 4 * F * F * n_seq * T / t / 1e9 / 157.3, n_seq * T / t / 1e3))
    del X, seqs
