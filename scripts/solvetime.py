"""tICA._solve wall time at the bench width (F = 512, k = 10) for the solve routes (host dsygvx; device finalise +
Cholesky reduction with host dsyevr; subspace iteration on the device; rocSOLVER), plus msm_potrf on its own."""
import ctypes as C, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA, _lib
warnings.simplefilter("ignore")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 512
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
g = torch.Generator(device="cuda"); g.manual_seed(0)
z = torch.cumsum(torch.randn(200000, 16, generator=g, device="cuda"), 0) * 0.01
X = (torch.tanh(z - z.mean(0)) @ torch.randn(16, F, generator=g, device="cuda") + 0.5 * torch.randn(200000, F, generator=g, device="cuda")).float()
seqs = list(X.view(20, 10000, F).unbind(0))
m = tICA(n_components=k, lag_time=100).fit(seqs)
def solve_ms(env, n=20):
    for a, b in env.items():
        os.environ[a] = b
    ts = []
    for _ in range(n + 3):
        m._is_dirty = True
        torch.cuda.synchronize(); t = time.perf_counter(); ev = m.eigenvalues_; ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts[3:])), ev.copy()
base = None
for name, env in (("host dsygvx", {"MSMBUILDER_AMD_DEVICE_SOLVE": "0"}),
                  ("device tail (subspace iteration, LAPACK fallback)", {"MSMBUILDER_AMD_DEVICE_SOLVE": "hybrid"}),
                  ("rocSOLVER dsyevd", {"MSMBUILDER_AMD_DEVICE_SOLVE": "1"})):
    ms, ev = solve_ms(env)
    print("      route:", getattr(m, "_solve_route", None))
    base = ev if base is None else base
    print("%-50s %.2f ms   max rel diff vs host %.1e" % (name, ms, np.abs(ev / base - 1).max()))
L = _lib.lib()
rs = np.random.RandomState(0)
M = rs.randn(F, F + 8); B = M @ M.T / F + 0.1 * np.eye(F)
dB = torch.from_numpy(B).cuda(); info = C.c_int()
def dev_ms(f, n=20):
    ts = []
    for _ in range(n + 3):
        torch.cuda.synchronize(); t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts[3:]))
w = dB.clone()
print("msm_potrf (device buffer, incl. sync)      %.3f ms" % dev_ms(lambda: L.msm_potrf(C.c_void_p(w.copy_(dB).data_ptr()), F, C.byref(info), 1)))
print("torch.linalg.cholesky                      %.3f ms" % dev_ms(lambda: (torch.linalg.cholesky(dB), torch.cuda.synchronize())))
