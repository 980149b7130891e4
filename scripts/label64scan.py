"""Profiling helper: the small-batch label kernel (B = 1024, K = 1000) over row widths, through msm_mbk_label."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import _lib
L = _lib.lib()
B, K = 1024, 1000
for m in (64, 128, 256, 512, 1024):
    h = C.c_void_p()
    cen = np.random.RandomState(0).randn(K, m).astype(np.float32)
    cnt = np.ones(K, np.float32)
    assert L.msm_mbk_create(C.byref(h), K, m) == 0
    assert L.msm_mbk_set(h, cen.ctypes.data, cnt.ctypes.data) == 0
    X = torch.randn(B, m, device="cuda")
    lab = torch.empty(B, dtype=torch.int32, device="cuda")
    for _ in range(20):
        assert L.msm_mbk_label(h, C.c_void_p(X.data_ptr()), B, C.c_void_p(lab.data_ptr()), None, 1) == 0
    torch.cuda.synchronize()
    L.msm_mbk_destroy(h)
