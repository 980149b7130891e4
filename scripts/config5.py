"""Profiling helper: BASELINE configs[4] width (F = 2048) and the bench width: fp32 vs bf16x2 vs bf16, float32 and
bfloat16-stored input; accumulate time = the pack + multiply pipeline (HIP events)."""
import ctypes as C, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA, _lib
warnings.simplefilter("ignore")
T, lag = 10000, 100
for F, n_seq in ((2048, 100), (512, 400)):
    X = torch.randn(n_seq * T, F, device="cuda") + 2.0
    Xb = X.to(torch.bfloat16)
    ref = None
    for mode, inp in (("f32", X), ("bf16x2", X), ("bf16", X), ("bf16x2", Xb), ("bf16", Xb)):
        os.environ["MSMBUILDER_AMD_TICA_MODE"] = mode
        seqs = list(inp.view(n_seq, T, F).unbind(0))
        best = 1e9
        for it in range(3):
            m = tICA(n_components=5, lag_time=lag).fit(seqs)
            ms = C.c_float(); _lib.check(_lib.lib().msm_tica_last_kernel_ms(m._handle, C.byref(ms)))
            if ms.value < best: best = ms.value
        ev = m.eigenvalues_
        if ref is None: ref = ev
        nt = (F + 127) // 128; nt2 = (F + 255) // 256
        exe = 2.0 * 128 * 128 * nt * (nt + 1) if mode == "f32" else 2.0 * 256 * 256 * nt2 * (nt2 + 1) * (4 if mode == "bf16x2" else 1)
        print("F=%d %-6s input %-8s: accumulate %.2f ms  %.1fM frames/s  executed %.0f TF (%.3f of %s peak)  alg %.0f TF  eig rel diff vs f32 %.1e" % (
            F, mode, str(inp.dtype).replace("torch.", ""), best, n_seq * T / best / 1e3, exe * n_seq * T / best / 1e9,
            exe * n_seq * T / best / 1e9 / (157.3 if mode == "f32" else 2500.0), "fp32" if mode == "f32" else "bf16",
            4.0 * F * F * n_seq * T / best / 1e9, np.abs(ev / ref - 1).max()))
    del X, Xb
