"""scripts/solvesweep.py -- randomized sweep of the hybrid solve against the host route (dsygvx): spectra with 0 .. 12 slow
processes, different noise levels, widths and component counts; prints the route taken and the worst disagreement."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from msmbuilder_amd import tICA
warnings.simplefilter("ignore")
os.environ["MSMBUILDER_AMD_TICA_MODE"] = "f64"
rs = np.random.RandomState(0)
bad = 0
routes = {}
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 48):
    F = int(rs.choice([256, 300, 512, 640, 1024]))
    n_slow = int(rs.choice([0, 1, 2, 3, 5, 8, 12]))
    k = int(rs.choice([2, 5, 10, 16]))
    noise = float(rs.choice([0.1, 0.5, 2.0]))
    n = int(rs.choice([3000, 8000, 20000]))
    lag = int(rs.choice([1, 5, 20]))
    osc = rs.rand() < 0.3
    z = np.zeros((n, max(n_slow, 1)))
    if n_slow:
        ts = np.logspace(np.log10(10.0), np.log10(800.0), n_slow)
        a = np.exp(-1.0 / ts)
        e = rs.randn(n, n_slow) * np.sqrt(1 - a * a)
        for t in range(1, n):
            z[t] = a * z[t - 1] + e[t]
    X = (z.dot(rs.randn(z.shape[1], F) / np.sqrt(z.shape[1])) if n_slow else 0) + noise * rs.randn(n, F) + rs.randn(F)
    if osc:
        X[:, 3] += 3.0 * np.cos(np.pi * 0.9 / lag * np.arange(n))
    seqs = [X[: n // 2], X[n // 2:]]
    out = {}
    for name, env in (("host", "0"), ("hybrid", "hybrid")):
        os.environ["MSMBUILDER_AMD_DEVICE_SOLVE"] = env
        try:
            m = tICA(n_components=k, lag_time=lag).fit(seqs)
            t = time.perf_counter(); ev = m.eigenvalues_.copy(); t = time.perf_counter() - t
            out[name] = (ev, m.eigenvectors_.copy(), m.covariance_.copy(), getattr(m, "_solve_route", None), t)
        except Exception as ex:
            out[name] = ex
    if isinstance(out["host"], Exception) or isinstance(out["hybrid"], Exception):
        same = type(out["host"]) is type(out["hybrid"])
        print("trial %2d F=%4d slow=%2d k=%2d noise=%.1f n=%5d lag=%2d osc=%d: exceptions %s / %s %s" % (trial, F, n_slow, k, noise, n, lag, osc, type(out["host"]).__name__, type(out["hybrid"]).__name__, "" if same else "  <-- DIFFERENT"))
        bad += 0 if same else 1
        continue
    ev, evh = out["hybrid"][0], out["host"][0]
    derr = np.abs(ev - evh).max()
    V, Vh, S = out["hybrid"][1], out["host"][1], out["host"][2]
    P = V.T.dot(S).dot(Vh)
    perr = np.abs(P.dot(P.T) - np.eye(k)).max()
    route = out["hybrid"][3]
    routes[route[0]] = routes.get(route[0], 0) + 1
    ok = derr <= 1e-10 and (perr <= 1e-5 or np.min(np.abs(np.diff(evh))) < 1e-6)
    bad += 0 if ok else 1
    print("trial %2d F=%4d slow=%2d k=%2d noise=%.1f n=%5d lag=%2d osc=%d: route %-22s solve %6.2f ms (host route %6.2f)  |d lambda| %.1e  span %.1e %s"
          % (trial, F, n_slow, k, noise, n, lag, osc, route, 1e3 * out["hybrid"][4], 1e3 * out["host"][4], derr, perr, "" if ok else "  <-- MISMATCH"))
print("routes", routes, "mismatches", bad)
