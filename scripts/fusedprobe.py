"""scripts/fusedprobe.py -- bf16 modes on bfloat16-STORED rows: the packed-image pipeline (tica_img_kernel + tica_img_pp_kernel) against
the fused kernel (tica_img_fused_kernel, MSM_TICA_IMG_FUSED=1) over feature widths (whole 256-feature panels), fit wall time and the
accumulation pipeline's HIP-event time."""
import ctypes as C, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA, _lib
warnings.simplefilter("ignore")
T, lag = 10000, 100
for F in (256, 512, 768, 1024, 1280, 1536, 2048):
    n_seq = max(20, int(2_000_000_000 // (F * 2 * T)))
    n_seq = min(n_seq, 200)
    X = (torch.randn(n_seq * T, F, device="cuda") + 1.0).to(torch.bfloat16)
    seqs = list(X.view(n_seq, T, F).unbind(0))
    for mode in ("bf16", "bf16x2"):
        os.environ["MSMBUILDER_AMD_TICA_MODE"] = mode
        res = {}
        for fused in ("0", "1"):
            os.environ["MSM_TICA_IMG_FUSED"] = fused
            ks, ws = [], []
            for it in range(4):
                torch.cuda.synchronize(); t = time.perf_counter()
                m = tICA(n_components=5, lag_time=lag).fit(seqs)
                torch.cuda.synchronize(); ws.append(time.perf_counter() - t)
                ms = C.c_float(); _lib.check(_lib.lib().msm_tica_last_kernel_ms(m._handle, C.byref(ms))); ks.append(ms.value)
            fl = C.c_int(0); _lib.check(_lib.lib().msm_tica_last_img_fused(m._handle, C.byref(fl)))
            res[fused] = (min(ks[1:]), 1e3 * min(ws[1:]), fl.value)
        print("F=%4d %-6s %4.1fM frames: image pipeline %6.2f ms (fit %6.2f)   fused %6.2f ms (fit %6.2f, ran fused: %d)   fused/image fit %.2f"
              % (F, mode, n_seq * T / 1e6, res["0"][0], res["0"][1], res["1"][0], res["1"][1], res["1"][2], res["1"][1] / res["0"][1]))
    del X, seqs
