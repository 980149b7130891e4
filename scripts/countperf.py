"""Profiling helper: transition-count throughput on device-resident labels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd.msm import _transition_counts

def timeit(fn, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) * 1e3)
    return best

for n_seq, T, K, lag, stay_p in ((1000, 10000, 200, 100, 0.98), (1000, 100000, 1000, 100, 0.98), (1000, 100000, 1000, 100, 0.0)):
    stay = torch.rand(n_seq, T, device="cuda") < stay_p
    draws = torch.randint(0, K, (n_seq, T), device="cuda")
    idx = torch.arange(T, device="cuda").expand(n_seq, T)
    last = torch.cummax(torch.where(~stay, idx, torch.zeros_like(idx)), dim=1).values
    labels = torch.gather(draws, 1, last)
    seqs = list(labels.unbind(0))
    del stay, draws, idx, last
    t = timeit(lambda: _transition_counts(seqs, lag_time=lag))
    n = n_seq * T
    print("%d labels, K=%d, lag=%d, stay=%.2f: %.2f ms  %.1fM frames/s  %.2f TB/s (3 passes x 8 B + 8 B)" % (
        n, K, lag, stay_p, t, n / t / 1e3, n * 32 / t / 1e9))
    del labels, seqs
