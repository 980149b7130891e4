"""Profiling helper: BASELINE config 5 width (F = 2048) -- bf16-MFMA covariance vs fp32, kernel time per mode."""
import ctypes as C, os, subprocess, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from msmbuilder_amd import tICA, _lib
    F, T, n_seq, lag = 2048, 10000, 100, 100
    torch.manual_seed(0)
    Z = torch.randn(n_seq * T, 16, device="cuda").cumsum(0) * 0.01
    X = Z @ torch.randn(16, F, device="cuda") + torch.randn(n_seq * T, F, device="cuda")
    del Z
    seqs = list(X.view(n_seq, T, F).unbind(0))
    ts = []
    for it in range(3):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = tICA(n_components=10, lag_time=lag).fit(seqs)
        ms = C.c_float(); _lib.check(_lib.lib().msm_tica_last_kernel_ms(m._handle, C.byref(ms))); ts.append(ms.value)
    t = min(ts)
    ev = m.eigenvalues_
    mode = os.environ["MSMBUILDER_AMD_TICA_MODE"]
    peak = {"f32": 157.3, "f64": 78.6}.get(mode, 2500.0)
    tf = 4.0 * F * F * n_seq * T / t / 1e9
    gbs = n_seq * T * F * 4 / t / 1e6
    print("mode %-6s F=%d N=%d: kernel %8.2f ms  %7.1f TF alg (%.3f of %s peak)  %6.2fM frames/s  input stream %.0f GB/s  top eig %s" % (
        mode, F, n_seq * T, t, tf, tf / peak, mode if mode in ("f32", "f64") else "bf16", n_seq * T / t / 1e3, gbs, np.round(ev[:3], 6)))
else:
    for mode in ("f32", "bf16x2", "bf16"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, MSMBUILDER_AMD_TICA_MODE=mode))
