"""cProfile of tICA.fit + eigenvalues_ on the bench shape (host-side overhead around the kernels)."""
import cProfile, os, pstats, sys, warnings, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmbuilder_amd import tICA
F, T, n_seq = 512, 10000, 1000
X = torch.randn(n_seq * T, F, device="cuda")
seqs = list(X.view(n_seq, T, F).unbind(0))
warnings.simplefilter("ignore")
if os.environ.get("SETSTREAM"):
    from msmbuilder_amd import _lib
    _lib.ensure_device(0); _lib.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(2):
    m = tICA(n_components=10, lag_time=100).fit(seqs); m.eigenvalues_
torch.cuda.synchronize()
t = time.perf_counter(); m = tICA(n_components=10, lag_time=100).fit(seqs); torch.cuda.synchronize(); t1 = time.perf_counter(); m.eigenvalues_; t2 = time.perf_counter()
print("fit %.2f ms, solve %.2f ms" % (1e3 * (t1 - t), 1e3 * (t2 - t1)))
for _ in range(3):
    t = time.perf_counter(); m = tICA(n_components=10, lag_time=100).fit(seqs); torch.cuda.synchronize(); t1 = time.perf_counter(); m.eigenvalues_; t2 = time.perf_counter()
    print("fit %.2f ms, solve %.2f ms" % (1e3 * (t1 - t), 1e3 * (t2 - t1)))
pr = cProfile.Profile(); pr.enable()
m = tICA(n_components=10, lag_time=100).fit(seqs); m.eigenvalues_
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
