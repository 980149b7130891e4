"""Profiling helper: host-side variants of the top-k generalized eigensolve at F = 512."""
import time, numpy as np, scipy.linalg
from threadpoolctl import ThreadpoolController
ctl = ThreadpoolController()
rs = np.random.RandomState(0)
F, k = 512, 10
A = rs.randn(F, 4 * F); S = A.dot(A.T) / (4 * F)
B = rs.randn(F, F); OC = 0.5 * S + 0.05 * (B + B.T)

def gvx():
    return scipy.linalg.eigh(OC, b=S, subset_by_index=[F - k, F - 1])
def chol(driver):
    def f():
        L = scipy.linalg.cholesky(S, lower=True, check_finite=False)
        Z = scipy.linalg.solve_triangular(L, OC, lower=True, check_finite=False)
        C = scipy.linalg.solve_triangular(L, Z.T, lower=True, check_finite=False)
        w, y = scipy.linalg.eigh(C, subset_by_index=[F - k, F - 1], driver=driver, check_finite=False)
        v = scipy.linalg.solve_triangular(L, y, lower=True, trans='T', check_finite=False)
        return w, v
    return f
w0, v0 = gvx()
for threads in (1, 2, 4, 8):
    with ctl.limit(limits=threads, user_api="blas"):
        for name, f in (("gvx", gvx), ("chol+evr", chol("evr")), ("chol+evx", chol("evx")), ("chol+evd(all)", None)):
            if f is None:
                continue
            f()
            t = time.perf_counter()
            for _ in range(5):
                w, v = f()
            dt = (time.perf_counter() - t) / 5 * 1e3
            print("threads=%d %-10s %.2f ms  max|dw| %.1e" % (threads, name, dt, np.abs(w - w0).max()))
