"""Round 6: timing ablations of the carried pack (tica_img_dev.h): 1M x 2048 bfloat16-stored rows, mode bf16, accumulate ms
(HIP events) with MSM_TICA_IMG_CARRY_ABL = 0 (everything), 1 (no pieces issued), 2 (no conversion), 3 (neither: the state
machine alone), and MSM_TICA_IMG_CARRY = 0 / 2 (pre-pass kernel; the carried pack's super-chunks packed by the pre-pass)."""
import ctypes as C, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA, _lib
warnings.simplefilter("ignore")
T, lag = 10000, 100
F = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
n_seq = 100 * 2048 // F
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
stored = sys.argv[2] if len(sys.argv) > 2 else "bfloat16"
X = torch.randn(n_seq * T, F, device="cuda") + 2.0
if stored == "bfloat16":
    X = X.to(torch.bfloat16)
os.environ["MSMBUILDER_AMD_TICA_MODE"] = mode
seqs = list(X.view(n_seq, T, F).unbind(0))
for carry, abl, fold in (("1", "0", "1"), ("1", "0", "0"), ("1", "1", "0"), ("1", "2", "0"), ("0", "0", "0"), ("0", "0", "1")):
    os.environ["MSM_TICA_IMG_CARRY"] = carry
    os.environ["MSM_TICA_IMG_CARRY_ABL"] = abl
    os.environ["MSM_TICA_FOLD"] = fold     # (0: a column-sum pass ahead, outside the timed events; the ablations' garbage never reaches a finite check)
    best = 1e9
    for it in range(4):
        m = tICA(n_components=5, lag_time=lag).fit(seqs)
        ms = C.c_float(); _lib.check(_lib.lib().msm_tica_last_kernel_ms(m._handle, C.byref(ms)))
        best = min(best, ms.value)
    print("F=%d " % F + "%s %s rows  MSM_TICA_IMG_CARRY=%s ABL=%s FOLD=%s : accumulate %.2f ms" % (mode, stored, carry, abl, fold, best))
