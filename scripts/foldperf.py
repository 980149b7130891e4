"""tICA.fit at 10M x 512 with the column sums folded into the sum/difference kernel (default) and with the separate pass
(MSM_TICA_FOLD=0), interleaved: wall time of fit and HIP-event time of the MFMA kernel, per call."""
import ctypes as C, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmbuilder_amd import tICA, _lib
F, T, n_seq = 512, 10000, int(os.environ.get("N_SEQ", "1000"))
X = torch.randn(n_seq * T, F, device="cuda")
X += torch.rand(F, device="cuda") * 2 - 1
seqs = list(X.view(n_seq, T, F).unbind(0))
warnings.simplefilter("ignore")
def one():
    torch.cuda.synchronize(); t = time.perf_counter()
    m = tICA(n_components=10, lag_time=100).fit(seqs); torch.cuda.synchronize(); t1 = time.perf_counter()
    ms = C.c_float(0); _lib.check(_lib.lib().msm_tica_last_kernel_ms(m._handle, C.byref(ms)))
    return 1e3 * (t1 - t), ms.value
res = {"1": [], "0": []}
envs = sys.argv[1:] or ["1", "0"]
for rep in range(6):
    for f in envs:
        os.environ["MSM_TICA_FOLD"] = f
        r = one()
        if rep: res.setdefault(f, []).append(r)
for f in envs:
    print("FOLD=%s  fit %s | kernel %s" % (f, " ".join("%.2f" % a for a, _ in res[f]), " ".join("%.2f" % b for _, b in res[f])))
