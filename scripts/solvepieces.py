"""Profiling helper: where a 512 x 512 generalized symmetric top-k solve spends its time, host vs device pieces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.linalg, torch
from threadpoolctl import threadpool_limits
F, k = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 10
rs = np.random.RandomState(0)
Z = rs.randn(20000, F) @ rs.randn(F, F) * 0.1
X0, X1 = Z[:-7], Z[7:]
B = (X0.T @ X0 + X1.T @ X1) / (2 * len(X0)); A = (X0.T @ X1 + X1.T @ X0) / (2 * len(X0))
def tm(f, n=5):
    f(); ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return 1e3 * min(ts)
for nt in (1, 4, 8):
    with threadpool_limits(nt, "blas"):
        t_gvx = tm(lambda: scipy.linalg.eigh(A, b=B, subset_by_index=[F - k, F - 1]))
        def chol_evr():
            L = scipy.linalg.cholesky(B, lower=True, check_finite=False)
            W = scipy.linalg.solve_triangular(L, A, lower=True, check_finite=False)
            Cs = scipy.linalg.solve_triangular(L, W.T, lower=True, check_finite=False)
            w, y = scipy.linalg.eigh(Cs, subset_by_index=[F - k, F - 1], driver="evr", check_finite=False)
            return w, scipy.linalg.solve_triangular(L, y, lower=True, trans="T", check_finite=False)
        t_ce = tm(chol_evr)
        L = scipy.linalg.cholesky(B, lower=True); W = scipy.linalg.solve_triangular(L, A, lower=True); Cs = scipy.linalg.solve_triangular(L, W.T, lower=True)
        Cs = 0.5 * (Cs + Cs.T)
        t_evr = tm(lambda: scipy.linalg.eigh(Cs, subset_by_index=[F - k, F - 1], driver="evr", check_finite=False))
        t_evx = tm(lambda: scipy.linalg.eigh(Cs, subset_by_index=[F - k, F - 1], driver="evx", check_finite=False))
        t_evd = tm(lambda: scipy.linalg.eigh(Cs, driver="evd", check_finite=False))
        t_trd = tm(lambda: scipy.linalg.lapack.dsytrd(Cs, lower=1))
        print("host F=%d threads=%d: dsygvx %.2f ms | chol+2trsm+evr+trsm %.2f | evr alone %.2f evx %.2f evd(all) %.2f sytrd %.2f" % (F, nt, t_gvx, t_ce, t_evr, t_evx, t_evd, t_trd))
w0 = scipy.linalg.eigh(A, b=B, subset_by_index=[F - k, F - 1])[0]
w1 = chol_evr()[0]
print("evr vs gvx eigenvalue max rel diff %.2e" % np.abs(w1 / w0 - 1).max())
dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
def dt(f, n=5):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return 1e3 * min(ts)
Ld = torch.linalg.cholesky(dB)
print("device: cholesky %.2f ms" % dt(lambda: torch.linalg.cholesky(dB)))
print("device: 2 x solve_triangular %.2f ms" % dt(lambda: torch.linalg.solve_triangular(Ld, torch.linalg.solve_triangular(Ld, dA, upper=False).T, upper=False)))
Csd = torch.from_numpy(Cs).cuda()
print("device: eigh (syevd all) %.2f ms" % dt(lambda: torch.linalg.eigh(Csd)))
print("device: eigvalsh %.2f ms" % dt(lambda: torch.linalg.eigvalsh(Csd)))
h = torch.empty(F, F, dtype=torch.float64).pin_memory()
print("D2H F*F f64 pinned %.3f ms, H2D %.3f ms" % (dt(lambda: h.copy_(Csd)), dt(lambda: Csd.copy_(h))))
print("device matmul FxF %.3f ms" % dt(lambda: Csd @ Csd))
# subspace: device gemm of F x F by F x 32
V = torch.randn(F, 32, dtype=torch.float64, device="cuda")
print("device FxF @ Fx32 %.3f ms; QR Fx32 %.3f ms" % (dt(lambda: Csd @ V), dt(lambda: torch.linalg.qr(V))))
