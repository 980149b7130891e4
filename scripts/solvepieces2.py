"""Profiling helper: wall time of the pieces of the hybrid solve with the device tridiagonalisation (F = 512)."""
import ctypes as C, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA, _lib
from msmbuilder_amd.decomposition import _moments
warnings.simplefilter("ignore")
F, k = 512, 10
X = torch.randn(200000, 16, device="cuda") @ torch.randn(16, F, device="cuda") + 0.5 * torch.randn(200000, F, device="cuda") + 3.0
X[1:] = 0.7 * X[:-1] + 0.3 * X[1:]
m = tICA(n_components=k, lag_time=100).fit(list(X.view(20, 10000, F).unbind(0)))
L = _lib.lib()
d, e, tau, Vr, Cs, mu, info = np.empty(F), np.empty(F - 1), np.empty(F - 1), np.empty((F - 1) ** 2), np.empty((F, F)), np.empty(F), np.zeros(8)
st = C.c_int(0)
def t(f, n=8):
    f(); torch.cuda.synchronize(); best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return 1e3 * best
t_red = t(lambda: L.msm_tica_reduce(m._handle, -1.0, m.n_observations_, None, Cs.ctypes.data, mu.ctypes.data, info.ctypes.data))
t_trd = t(lambda: L.msm_tica_reduce_tridiag(m._handle, -1.0, m.n_observations_, None, d.ctypes.data, e.ctypes.data, tau.ctypes.data, Vr.ctypes.data, Cs.ctypes.data, mu.ctypes.data, info.ctypes.data, C.byref(st)))
t_host = t(lambda: _moments.eigenpairs_from_tridiagonal(d, e, tau, Vr, k))
vals, Y = _moments.eigenpairs_from_tridiagonal(d, e, tau, Vr, k)
V = np.empty((k, F))
t_back = t(lambda: L.msm_tica_backsolve(m._handle, Y.ctypes.data, k, V.ctypes.data))
import scipy.linalg
t_stemr = t(lambda: scipy.linalg.eigh_tridiagonal(d, e, select='i', select_range=(F - k, F - 1), lapack_driver='stemr', check_finite=False))
print("reduce (export+finalise+potrf+2 trsm+D2H Cs) %.2f ms | reduce+sytrd+D2H(V,d,e,tau) %.2f ms | host stemr+ormqr %.2f ms (stemr alone %.2f) | backsolve %.2f ms | status %d" % (t_red, t_trd, t_host, t_stemr, t_back, st.value))
