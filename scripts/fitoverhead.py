"""Host-side cost of tICA.fit per trajectory: the same 10M x 512 frames as 1,000 / 100 / 10 trajectories (and as one 3-D
tensor), wall time of fit minus the HIP-event time of the MFMA kernel."""
import ctypes as C, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmbuilder_amd import tICA, _lib
warnings.simplefilter("ignore")
F, N = 512, 10_000_000
X = torch.randn(N, F, device="cuda")
def run(seqs, label):
    res = []
    for rep in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        m = tICA(n_components=10, lag_time=100).fit(seqs); torch.cuda.synchronize(); w = 1e3 * (time.perf_counter() - t)
        ms = C.c_float(0); _lib.check(_lib.lib().msm_tica_last_kernel_ms(m._handle, C.byref(ms)))
        if rep: res.append((w, ms.value))
    print("%-28s fit %s | fit - kernel %s" % (label, " ".join("%.2f" % a for a, _ in res), " ".join("%.2f" % (a - b) for a, b in res)))
for n_seq in (1000, 100, 10):
    run(list(X.view(n_seq, N // n_seq, F).unbind(0)), "%d trajectories (list)" % n_seq)
run(X.view(1000, N // 1000, F), "1000 trajectories (3-D tensor)")
