"""Experiment (round 5): the bf16 image path's packing pre-pass against chunk size (MSM_TICA_IMG_KC) -- run once per setting
(the value is read once per process): accumulate ms of bf16 / bf16x2 fits of 1M x 2048, float32 and bfloat16-stored rows."""
import ctypes as C, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA, _lib
warnings.simplefilter("ignore")
T, lag, F, n_seq = 10000, 100, 2048, 100
X = torch.randn(n_seq * T, F, device="cuda") + 2.0
Xb = X.to(torch.bfloat16)
for fused in ("0", "1"):
    os.environ["MSM_TICA_IMG_FUSED"] = fused
    for mode, inp in (("bf16", X), ("bf16", Xb), ("bf16x2", Xb)):
        if fused == "1" and inp is X:
            continue
        os.environ["MSMBUILDER_AMD_TICA_MODE"] = mode
        seqs = list(inp.view(n_seq, T, F).unbind(0))
        best, wall = 1e9, 1e9
        import time
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m = tICA(n_components=5, lag_time=lag).fit(seqs)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            ms = C.c_float(); _lib.check(_lib.lib().msm_tica_last_kernel_ms(m._handle, C.byref(ms)))
            best = min(best, ms.value); wall = min(wall, 1e3 * (t1 - t0))
        print("KC=%s RING=%s fused=%s %-6s input %-8s: accumulate %.2f ms, fit wall %.2f ms" % (os.environ.get("MSM_TICA_IMG_KC", "-"),
              os.environ.get("MSM_TICA_IMG_RING_MB", "-"), fused, mode, str(inp.dtype).replace("torch.", ""), best, wall))
