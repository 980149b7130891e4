import time, numpy as np, torch, scipy.linalg
torch.manual_seed(0)
for F in (512, 1024, 2048):
    A = torch.randn(F, 4 * F, dtype=torch.float64, device="cuda"); S = (A @ A.T) / (4 * F)
    B = torch.randn(F, F, dtype=torch.float64, device="cuda"); OC = (B + B.T) * 0.05 + 0.5 * S
    def gpu():
        L = torch.linalg.cholesky(S)
        Z = torch.linalg.solve_triangular(L, OC, upper=False)
        M = torch.linalg.solve_triangular(L, Z.T, upper=False)
        M = (M + M.T) * 0.5
        w, V = torch.linalg.eigh(M)
        vec = torch.linalg.solve_triangular(L.T, V[:, -10:], upper=True)
        return w[-10:].cpu(), vec.cpu()
    for _ in range(2): gpu()
    torch.cuda.synchronize(); t = time.perf_counter(); w, v = gpu(); torch.cuda.synchronize(); tg = time.perf_counter() - t
    Sh, Oh = S.cpu().numpy(), OC.cpu().numpy()
    from threadpoolctl import threadpool_limits
    with threadpool_limits(1 if F <= 768 else 8, user_api="blas"):
        t = time.perf_counter(); wr, vr = scipy.linalg.eigh(Oh, b=Sh, subset_by_index=[F - 10, F - 1]); tc = time.perf_counter() - t
    print("F=%d  gpu(torch) %.1f ms   cpu(scipy) %.1f ms   max|dw| %.2e" % (F, tg * 1e3, tc * 1e3, np.abs(w.numpy() - wr).max()))
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from msmbuilder_amd.decomposition import _moments
    for _ in range(2):
        t = time.perf_counter(); wd, vd = _moments.device_generalized_eigenpairs(Oh, Sh, 10); tn = time.perf_counter() - t
    print("F=%d  msm_sygv_top (host arrays in, top-10 out) %.1f ms   max|dw| %.2e" % (F, tn * 1e3, np.abs(wd[::-1] - wr).max()))
