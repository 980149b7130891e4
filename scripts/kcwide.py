"""scripts/kcwide.py -- KCenters.fit on the shapes the wide screened passes (csrc/distance_wscreen_dev.h) are for, with the
screen on and off (MSM_KC_WSCREEN is read per fit), wall ms (min of 3) + the pass counts of msm_kcenters_last_stats."""
import ctypes as C, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import KCenters, _lib
warnings.simplefilter("ignore")
g = torch.Generator(device="cuda").manual_seed(5)
def data(n, m, kc, dtype):
    cen = torch.randn(kc, m, generator=g, device="cuda") * 3
    return (cen[torch.randint(0, kc, (n,), generator=g, device="cuda")] + torch.randn(n, m, generator=g, device="cuda")).to(dtype).contiguous()
def fit_ms(X, K):
    f = lambda: KCenters(n_clusters=K, random_state=0).fit([X])
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    st = (C.c_int64 * 5)(); _lib.check(_lib.lib().msm_kcenters_last_stats(st))
    os.environ["MSM_KC_STATS"] = "1"; f(); os.environ["MSM_KC_STATS"] = "0"
    ws = (C.c_int64 * 2)(); _lib.check(_lib.lib().msm_kcenters_last_wide_stats(ws))
    return 1e3 * min(ts), list(st) + list(ws)
shapes = [(280_000, 171, 200, torch.float32, "C3 stress: contact-like rows"), (1_000_000, 171, 500, torch.float32, ""), (1_000_000, 64, 200, torch.float32, ""),
          (500_000, 512, 200, torch.float32, ""), (2_000_000, 10, 200, torch.float32, ""), (2_000_000, 17, 200, torch.float64, ""),
          (2_000_000, 40, 200, torch.float64, ""), (2_000_000, 16, 200, torch.float64, "(register-resident screen: unchanged)")]
for n, m, K, dt, note in shapes:
    if n == 280_000:   # the bench leg's data: one blob of contact-like distances
        gC = torch.Generator(device="cuda").manual_seed(171)
        X = (torch.linspace(0.4, 2.5, 171, device="cuda") + 0.2 * torch.randn(n, 171, generator=gC, device="cuda")).abs().float().contiguous()
    else:
        X = data(n, m, 50, dt)
    os.environ["MSM_KC_WSCREEN"] = "1"
    os.environ["MSM_KC_WBATCH"] = "1"
    bat, stb = fit_ms(X, K)
    print("KCenters(%4d).fit %8d x %3d %-8s batched  %8.2f ms (%d plain + %d batched passes)" % (K, n, m, str(dt)[6:], bat, stb[1], stb[2]), flush=True)
    os.environ["MSM_KC_WBATCH"] = "0"
    on, st = fit_ms(X, K)
    os.environ["MSM_KC_WSCREEN"] = "0"
    off, _ = fit_ms(X, K)
    print("KCenters(%4d).fit %8d x %3d %-8s screened %8.2f ms (%d plain + %d screened passes, %d B/row screened; %.2f %% of rows re-evaluated per pass, %.2f %% changed) | plain passes %8.2f ms   %s"
          % (K, n, m, str(dt)[6:], on, st[1], st[2], st[4], 100.0 * st[5] / max(1, st[2]) / n, 100.0 * st[6] / max(1, st[2]) / n, off, note))
    del X
