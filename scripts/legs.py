"""One pass over every leg the bench prints, at moderate sizes, for the PMC passes of scripts/r04_pmc_all.sh (each leg runs
its kernels a known number of times; the table printed at the end gives the ALGORITHMIC bytes of each leg's dominant
kernel per launch, to set beside FETCH_SIZE x 2 + WRITE_SIZE).  `legs.py c5`: only BASELINE configs[4]'s width at
1,000,000 x 2048 bfloat16-stored rows, one fit per bf16 mode (scripts/r04_final.sh c5)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmbuilder_amd import tICA, KCenters, MiniBatchKMeans
warnings.simplefilter("ignore")
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(4)
T = 10_000
rows = []


def seqs_of(X, n):
    return list(X.view(n, T, X.shape[1]).unbind(0))


if len(sys.argv) > 1 and sys.argv[1] == "c5":
    Xb = (torch.randn(100 * T, 2048, generator=g, device=dev) + 1.0).to(torch.bfloat16)
    for mode in ("bf16", "bf16x2"):
        os.environ["MSMBUILDER_AMD_TICA_MODE"] = mode
        tICA(n_components=10, lag_time=100).fit(seqs_of(Xb, 100))
    torch.cuda.synchronize()
    print("ALGORITHMIC bytes per fit: %.4g B (1,000,000 x 2048 bfloat16-stored)" % (Xb.numel() * 2))
    sys.exit(0)


# 1) fp32 sum/difference kernel (the bench's dominant kernel), 2M x 512
X = torch.randn(200 * T, 512, generator=g, device=dev) + 1.0
os.environ["MSMBUILDER_AMD_TICA_MODE"] = "f32"
m32 = tICA(n_components=10, lag_time=100).fit(seqs_of(X, 200))
rows.append(("tica_sym_f32_kernel", "2M x 512 fp32 fit", X.numel() * 4))
# 2) fp64 kernel on the same data (first 1M frames)
os.environ["MSMBUILDER_AMD_TICA_MODE"] = "f64"
tICA(n_components=10, lag_time=100).fit(seqs_of(X[: 100 * T], 100))
rows.append(("tica_mfma_f64_kernel", "1M x 512 fp32 input, f64 mode", 100 * T * 512 * 4))
os.environ["MSMBUILDER_AMD_TICA_MODE"] = "f32"
# 3) projection
Y = m32.transform([X])[0]
rows.append(("tica_project_mfma_kernel", "2M x 512 fp32 -> 10 fp64", X.numel() * 4 + Y.numel() * 8))
# 4) k-centers on the projection (batched passes on the byte copy) + screened assign
kc = KCenters(n_clusters=200, random_state=0).fit([Y])
kc.predict([Y])
rows.append(("kcenters_batch_pass_kernel", "2M x 10 fp64, per pass: the 16-byte-per-row copy", Y.shape[0] * 16))
rows.append(("assign_screen_kernel", "2M x 10 fp64 rows, K = 200", Y.numel() * 8 + Y.shape[0] * 8))
del Y, kc
# 5) MiniBatchKMeans final labelling, 1M x 512, K = 1000
Xl = X[: 100 * T]
mb = MiniBatchKMeans(n_clusters=1000, random_state=0, max_iter=1, n_init=1).fit([Xl])
mb.predict([Xl])
rows.append(("kmeans_label_v4_kernel", "1M x 512 fp32, K = 1000, per labelling pass", Xl.numel() * 4))
del X, Xl, m32, mb
torch.cuda.empty_cache()
# 6) bf16 image path at BASELINE configs[4]'s width, bfloat16-STORED input, 500k x 2048
Xb = (torch.randn(50 * T, 2048, generator=g, device=dev) + 1.0).to(torch.bfloat16)
for mode in ("bf16", "bf16x2"):
    os.environ["MSMBUILDER_AMD_TICA_MODE"] = mode
    tICA(n_components=10, lag_time=100).fit(seqs_of(Xb, 50))
rows.append(("tica_img_kernel + tica_img_pp_kernel", "500k x 2048 bfloat16-stored, per fit (bf16 / bf16x2 fits: one each)", Xb.numel() * 2))
os.environ["MSMBUILDER_AMD_TICA_MODE"] = "f32"
del Xb
# 7) SURVEY 8(d) C3 stress variant: wide exact kernels
XC = (torch.linspace(0.4, 2.5, 171, device=dev) + 0.2 * torch.randn(280_000, 171, generator=g, device=dev)).abs().float().contiguous()
kcC = KCenters(n_clusters=200, random_state=0).fit(seqs_of(XC, 28))
kcC.predict(seqs_of(XC, 28))
rows.append(("kcenters_pass_kernel / pair_kernel (C3 stress)", "280,000 x 171 fp32, K = 200, per pass over X", XC.numel() * 4))
torch.cuda.synchronize()
print("ALGORITHMIC bytes per launch (or per fit where stated):")
for k, what, b in rows:
    print("  %-48s %-72s %.4g B" % (k, what, b))
