"""Profiling / validation helper: device Householder tridiagonalisation (msm_sytrd) against LAPACK, and tICA._solve with it."""
import ctypes as C, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.linalg, torch
from msmbuilder_amd import tICA, _lib
from msmbuilder_amd.decomposition import _moments
_lib.ensure_device(0)
L = _lib.lib()
rs = np.random.RandomState(0)
for n in (1, 2, 3, 17, 100, 256, 300, 512, 777, 1024):
    M = rs.randn(n, n); A = (M + M.T) / 2 + np.diag(rs.randn(n))
    d, e, tau, V = np.empty(n), np.empty(max(n - 1, 1)), np.empty(max(n - 1, 1)), np.empty(max(n - 1, 1) ** 2)
    st = C.c_int(0)
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        _lib.check(L.msm_sytrd(A.ctypes.data, n, d.ctypes.data, e.ctypes.data, tau.ctypes.data, V.ctypes.data, C.byref(st), 0))
        ts.append(time.perf_counter() - t)
    k = min(n, 10)
    w, Y = _moments.eigenpairs_from_tridiagonal(d, e[:n - 1], tau[:n - 1], V, k)
    wr, Zr = scipy.linalg.eigh(A, subset_by_index=[n - k, n - 1])
    wr, Zr = wr[::-1], Zr[:, ::-1]
    res = np.abs(A @ Y.T - Y.T * w).max()
    print("n=%4d status %d  sytrd+copies %.2f ms  eig max abs err %.1e  residual %.1e  orth %.1e" % (
        n, st.value, 1e3 * min(ts), np.abs(w - wr).max(), res, np.abs(Y @ Y.T - np.eye(k)).max()))
F = 512
X = torch.randn(200000, 16, device="cuda") @ torch.randn(16, F, device="cuda") + 0.5 * torch.randn(200000, F, device="cuda") + 3.0
X[1:] = 0.7 * X[:-1] + 0.3 * X[1:]
seqs = list(X.view(20, 10000, F).unbind(0))
res = {}
for name, env in (("hybrid/host-evr", {"MSMBUILDER_AMD_DEVICE_SOLVE": "hybrid", "MSMBUILDER_AMD_DEVICE_TRD": "0"}),
                  ("hybrid/device-trd", {"MSMBUILDER_AMD_DEVICE_SOLVE": "hybrid", "MSMBUILDER_AMD_DEVICE_TRD": "1"})):
    os.environ.update(env)
    ts = []
    for it in range(6):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = tICA(n_components=10, lag_time=100).fit(seqs)
            torch.cuda.synchronize(); t = time.perf_counter()
            ev = m.eigenvalues_; Vv = m.eigenvectors_
            ts.append(time.perf_counter() - t)
    res[name] = (ev.copy(), Vv.copy())
    print("F=%d %s: _solve %.2f ms  ev[:3] %s" % (F, name, 1e3 * min(ts[1:]), ev[:3]))
a, b = res["hybrid/host-evr"], res["hybrid/device-trd"]
sg = np.sign((a[1] * b[1]).sum(0))
print("device-trd vs host-evr: eig rel %.1e  vec abs %.1e" % (np.abs(b[0] / a[0] - 1).max(), np.abs(b[1] * sg - a[1]).max() / np.abs(a[1]).max()))
