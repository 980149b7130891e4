"""BASELINE.json configs that round 1 left untested (VERDICT r1 "close the test holes"):

* configs[0] shape -- 10 trajectories x 9,999 frames x 4 features (sin/cos of phi, psi), tICA(n_components=2,
  lag_time=1) -- f32 and f64 against the oracle;
* the BENCH shape itself, 10M x 512 fp32, lag 100: accumulators against an independent float64 contraction (torch, on the
  device), eigenvalues against a float64 solve of those moments at the stated rtol 1e-5;
* the sum/difference kernel's widths [512, lag 100] and [2048, lag 20]: eigenvalues against the ORACLE at rtol 1e-5 on
  well-sampled data (round 1 compared them with the other HIP kernel at atol 5e-5);
* MiniBatchKMeans(n_clusters=1000).fit on 100k x 512 against scikit-learn itself (n_steps_ equal, centres and inertia
  rtol 1e-4)."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fit(seqs, mode, monkeypatch, **kw):
    from msmbuilder_amd import tICA
    from oracle.tica_oracle import TicaOracle
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", mode)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return tICA(**kw).fit(seqs), TicaOracle(**kw).fit(seqs)


@pytest.mark.parametrize("mode,rtol", [("f32", 1e-5), ("f64", 1e-10)])
def test_config1_alanine_dipeptide_shape(gpu, monkeypatch, mode, rtol):
    """SURVEY 8(d) "C1-synthetic": two slowly diffusing dihedral angles -> [sin phi, cos phi, sin psi, cos psi]."""
    rs = np.random.RandomState(1234)
    seqs = []
    for _ in range(10):
        ang = np.cumsum(rs.randn(9999, 2) * 0.08, axis=0) + rs.uniform(-np.pi, np.pi, size=2)
        seqs.append(np.stack([np.sin(ang[:, 0]), np.cos(ang[:, 0]), np.sin(ang[:, 1]), np.cos(ang[:, 1])], axis=1).astype(np.float32))
    m, o = _fit(seqs, mode, monkeypatch, n_components=2, lag_time=1)
    assert (m.n_observations_, m.n_sequences_, m.n_features) == (99990, 10, 4)
    np.testing.assert_allclose(m.eigenvalues_, o.eigenvalues_, rtol=rtol)
    np.testing.assert_allclose(m.timescales_, o.timescales_, rtol=rtol * 200)      # -1 / ln(lambda), lambda ~ 0.99
    np.testing.assert_allclose(m.means_, o.means_, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(m.covariance_, o.covariance_, rtol=0, atol=rtol * np.abs(o.covariance_).max())
    np.testing.assert_allclose(m.offset_correlation_, o.offset_correlation_, rtol=0, atol=rtol * np.abs(o.offset_correlation_).max())
    Y, Yo = m.transform(seqs[:2]), o.transform(seqs[:2])
    for a, b in zip(Y, Yo):
        sign = np.sign((a * b).sum(0))
        np.testing.assert_allclose(a * sign, b, rtol=1e-3, atol=1e-4)
        assert a.shape == (9999, 2) and a.dtype == np.float64


def test_bench_shape_10M_x_512_against_fp64_contraction(gpu, monkeypatch):
    """The headline run's own shape and data recipe (bench.synth): 1,000 x 10,000 x 512 fp32, lag 100, default f32 mode
    (sum/difference kernel).  Reference: per-trajectory float64 matmuls in torch on the device (an independent
    contraction: different kernels, different summation order), finalised and solved on the host by the oracle's
    formulas."""
    import torch
    import scipy.linalg
    import bench
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f32")
    n_seq, T, F, lag, k = 1000, 10_000, 512, 100, 10
    X = bench.synth(torch, n_seq, T, F, 1234, torch.device("cuda"))
    seqs = list(X.view(n_seq, T, F).unbind(0))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = tICA(n_components=k, lag_time=lag).fit(seqs)
        ev = np.asarray(m.eigenvalues_)
    assert m._lagged_symmetrised and m.n_observations_ == n_seq * T
    C = torch.zeros(F, F, dtype=torch.float64, device="cuda")
    G = torch.zeros_like(C)
    s0 = torch.zeros(F, dtype=torch.float64, device="cuda")
    st = torch.zeros_like(s0)
    for s in range(0, n_seq, 20):
        V = X.view(n_seq, T, F)[s:s + 20].double()
        A, B = V[:, :-lag].reshape(-1, F), V[:, lag:].reshape(-1, F)
        C += A.T @ B
        G += A.T @ A + B.T @ B
        s0 += A.sum(0)
        st += B.sum(0)
        del V, A, B
    C, G, s0, st = C.cpu().numpy(), G.cpu().numpy(), s0.cpu().numpy(), st.cpu().numpy()
    m._pull()
    two_n = 2.0 * n_seq * (T - lag)
    mu = (s0 + st) / two_n
    S = G / two_n - np.outer(mu, mu)
    OC = (C + C.T) / two_n - np.outer(mu, mu)
    # accumulators: fp32-class error relative to the CENTRED moment (the mean shift), fp64 rounding of the raw totals
    atol = 1e-6 * two_n * np.abs(S).max() + 1e-11 * np.abs(G).max()
    np.testing.assert_allclose(m._outer_gram_sum, G, rtol=0, atol=atol)
    np.testing.assert_allclose(m._outer_0_to_T_lagged, 0.5 * (C + C.T), rtol=0, atol=atol)
    np.testing.assert_allclose(m._sum_0_to_TminusTau, s0, rtol=1e-11)
    np.testing.assert_allclose(m._sum_tau_to_T, st, rtol=1e-11)
    from oracle.tica_oracle import rao_blackwell_ledoit_wolf
    Sig, rho = rao_blackwell_ledoit_wolf(S, n_seq * T)
    ref = scipy.linalg.eigh(OC, b=Sig, subset_by_index=[F - k, F - 1])[0][::-1]
    np.testing.assert_allclose(ev, ref, rtol=1e-5)
    assert abs(m.shrinkage_ - rho) <= 1e-6 * rho
    np.testing.assert_allclose(m.covariance_, Sig, rtol=0, atol=1e-5 * np.abs(Sig).max())


@pytest.mark.parametrize("F,lag,n_seq,T", [(512, 100, 12, 12000), (2048, 20, 6, 10000)])
def test_sum_difference_kernel_eigenvalues_vs_oracle(gpu, monkeypatch, F, lag, n_seq, T):
    """Well-sampled (>= 25 frames per feature), slow modes much longer than the lag: the leading eigenvalues are
    separated, and the f32 default must reproduce the float64 oracle to the STATED rtol 1e-5."""
    rs = np.random.RandomState(F + lag)
    k = 8
    M = rs.randn(k, F) / np.sqrt(k)
    a = np.exp(-1.0 / (lag * np.array([40.0, 25.0, 15.0, 9.0, 5.0, 3.0, 2.0, 1.2])))
    seqs = []
    for _ in range(n_seq):
        e = rs.randn(T, k)
        z = np.empty((T, k))
        z[0] = e[0]
        for t in range(1, T):
            z[t] = a * z[t - 1] + np.sqrt(1 - a * a) * e[t]
        seqs.append((z @ M + 0.4 * rs.randn(T, F) + rs.uniform(-3, 3, size=F)).astype(np.float32))
    m, o = _fit(seqs, "f32", monkeypatch, n_components=4, lag_time=lag)
    assert m._lagged_symmetrised
    np.testing.assert_allclose(m.eigenvalues_, o.eigenvalues_, rtol=1e-5)
    np.testing.assert_allclose(m.covariance_, o.covariance_, rtol=0, atol=1e-5 * np.abs(o.covariance_).max())
    np.testing.assert_allclose(m.offset_correlation_, o.offset_correlation_, rtol=0, atol=1e-5 * np.abs(o.offset_correlation_).max())


@pytest.mark.parametrize("reassignment_ratio", [0.0, 0.01])
def test_config4_minibatchkmeans_k1000_fit_vs_sklearn(gpu, reassignment_ratio):
    """configs[3]'s clusterer at its K: MiniBatchKMeans(n_clusters=1000).fit on 100k x 512 against scikit-learn itself on
    the same init and the same RNG stream: equal step counts (same mini-batches, same early-stopping decisions), centres and
    inertia to rtol 1e-4, labels equal up to fp32 near-ties.  The data are 1000 separated blobs seeded with one member each, so
    that no label hangs on an fp32 near-tie (one flipped label changes a count, a count changes a reassignment draw, and
    from there the two RNG streams -- and fits -- legitimately diverge: that is sklearn's algorithm, not an error)."""
    sk = pytest.importorskip("sklearn.cluster")
    import torch
    from msmbuilder_amd import MiniBatchKMeans
    rs = np.random.RandomState(17)
    K, F, N = 1000, 512, 100_000
    cent = rs.randn(K, F).astype(np.float32) * 1.5
    member = rs.permutation(np.repeat(np.arange(K), N // K))
    X = (cent[member] + 0.5 * rs.randn(N, F).astype(np.float32)).astype(np.float32)
    first = np.array([np.nonzero(member == c)[0][0] for c in range(K)])
    init = X[first].copy()
    kw = dict(n_clusters=K, init=init, n_init=1, batch_size=1024, max_iter=2, random_state=5,
              reassignment_ratio=reassignment_ratio)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = sk.MiniBatchKMeans(**kw).fit(X)
        mine = MiniBatchKMeans(**kw).fit([torch.from_numpy(X).cuda()])
    assert mine.n_steps_ == ref.n_steps_
    np.testing.assert_allclose(mine.cluster_centers_, ref.cluster_centers_, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(mine.inertia_, ref.inertia_, rtol=1e-4)
    lab = mine.labels_[0].cpu().numpy()
    assert (lab != ref.labels_).mean() < 1e-3


@pytest.mark.parametrize("mode,rtol,ctol", [("bf16x2", 1e-5, 1e-5), ("bf16", 1e-3, 5e-3)])
@pytest.mark.parametrize("F,lag", [(2048, 20), (300, 7)])
def test_config5_bf16_image_path_vs_oracle(gpu, monkeypatch, mode, rtol, ctol, F, lag):
    """configs[4]'s width against the float64 ORACLE (round 1 compared bf16 only with the fp32 HIP path): the packed bf16
    sum/difference image + 256 x 256-tile MFMA kernel, un-centred features (the mean shift precedes the rounding), ragged
    trajectories, a width that is not a multiple of the 256-feature tiles, float32 and bfloat16-STORED input."""
    import torch
    rs = np.random.RandomState(F)
    k = 8
    M = rs.randn(k, F) / np.sqrt(k)
    a = np.exp(-1.0 / (lag * np.array([40.0, 25.0, 15.0, 9.0, 5.0, 3.0, 2.0, 1.2])))
    seqs = []
    for T in (12000, 9001, 4100 + lag, lag + 1, lag):
        e = rs.randn(T, k)
        z = np.empty((T, k))
        z[0] = e[0]
        for t in range(1, T):
            z[t] = a * z[t - 1] + np.sqrt(1 - a * a) * e[t]
        seqs.append((z @ M + 0.4 * rs.randn(T, F) + 8.0 * np.sign(rs.randn(F))).astype(np.float32))
    m, o = _fit(seqs, mode, monkeypatch, n_components=4, lag_time=lag)
    assert m._lagged_symmetrised and (m.n_observations_, m.n_sequences_) == (o.n_observations_, o.n_sequences_)
    np.testing.assert_allclose(m.eigenvalues_, o.eigenvalues_, rtol=rtol)
    np.testing.assert_allclose(m.means_, o.means_, rtol=1e-10)
    np.testing.assert_allclose(m.covariance_, o.covariance_, rtol=0, atol=ctol * np.abs(o.covariance_).max())
    np.testing.assert_allclose(m.offset_correlation_, o.offset_correlation_, rtol=0, atol=ctol * np.abs(o.offset_correlation_).max())
    # bfloat16-stored trajectories: the oracle sees exactly the stored values
    from msmbuilder_amd import tICA
    from oracle.tica_oracle import TicaOracle
    dev = [torch.from_numpy(x).cuda().to(torch.bfloat16) for x in seqs]
    host = [x.float().cpu().numpy() for x in dev]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mb = tICA(n_components=4, lag_time=lag).fit(dev)
        ob = TicaOracle(n_components=4, lag_time=lag).fit(host)
    np.testing.assert_allclose(mb.eigenvalues_, ob.eigenvalues_, rtol=rtol)
    np.testing.assert_allclose(mb.means_, ob.means_, rtol=1e-10)
    np.testing.assert_allclose(mb.covariance_, ob.covariance_, rtol=0, atol=ctol * np.abs(ob.covariance_).max())


@pytest.mark.parametrize("mode", ["bf16", "bf16x2"])
@pytest.mark.parametrize("F,lag", [(2048, 100), (512, 7), (256, 33)])
def test_config5_fused_kernel_bit_identical_to_image_path(gpu, monkeypatch, mode, F, lag):
    """Round 5: with MSM_TICA_IMG_FUSED=1 bfloat16-stored rows of whole 256-feature panels run the FUSED kernel (no packed
    image; the MFMA kernel's load role forms u = x_t + x_{t+tau} - 2 r and d = x_t - x_{t+tau} from the raw rows).  Its
    accumulators must equal the packed-image pipeline's bit for bit: same products, same order, same roundings.  Ragged
    trajectories: lengths that leave 1, 12, 17 and 31 pairs in the last K-step, one of lag + 1 rows, one too short."""
    import ctypes as C
    import torch
    from msmbuilder_amd import tICA, _lib
    g = torch.Generator(device="cuda").manual_seed(F + lag)
    lens = [3 * 32 + lag + 1, 40 * 32 + lag + 12, 2000 + lag + 17, 1500 * 2 + lag + 31 - 8, lag + 1, lag, 4096 + lag]
    base = 3.0 * torch.randn(F, generator=g, device="cuda")
    seqs = [(base + torch.randn(n, F, generator=g, device="cuda").cumsum(0) * 0.05
             + torch.randn(n, F, generator=g, device="cuda")).to(torch.bfloat16) for n in lens]
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", mode)
    monkeypatch.setenv("MSM_TICA_FOLD", "0")      # both paths take the same column-sum pass: every exported word comparable
    monkeypatch.setenv("MSM_TICA_IMG_CARRY", "0")  # one super-chunk, one launch, like the fused kernel's: the same fp32 partial sums (the carried pack cuts a launch into several)
    out = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("MSM_TICA_IMG_FUSED", fused)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = tICA(n_components=3, lag_time=lag).fit(seqs)
            # a second call on the same handle: partial_fit on top of existing slabs (the shift row is already set)
            m.partial_fit(seqs[1])
        flag = C.c_int(-1)
        _lib.check(_lib.lib().msm_tica_last_img_fused(m._handle, C.byref(flag)))
        assert flag.value == int(fused)
        m._pull()
        out[fused] = [np.array(getattr(m, a)) for a in ("_outer_0_to_T_lagged", "_outer_gram_sum", "_sum_0_to_TminusTau",
                                                         "_sum_tau_to_T")] + [np.asarray(m.eigenvalues_)]
        assert (m.n_observations_, m.n_sequences_) == (sum(n for n in lens if n > lag) + lens[1], 7)
        assert np.abs(out[fused][0]).max() > 0 and np.isfinite(out[fused][1]).all()
    for a, b in zip(out["1"], out["0"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("mode", ["bf16", "bf16x2"])
@pytest.mark.parametrize("stored,F,lag", [("bfloat16", 2048, 100), ("float32", 2048, 100), ("bfloat16", 768, 7), ("float32", 1536, 33)])
@pytest.mark.parametrize("fold", ["0", "2"])
def test_config5_carried_pack_bit_identical_to_prepass(gpu, monkeypatch, mode, stored, F, lag, fold):
    """Round 6 (VERDICT r5 #3b): the multiply of super-chunk k packs super-chunk k + 1 of the image in its load role (the
    carried pack, tica_img_dev.h); only the first super-chunk takes the pre-pass kernel.  The packets are the pre-pass
    kernel's bit for bit, so every accumulator is too.  Ragged trajectories (1, 12, 17 and 31 pairs in a last K-step, one
    of lag + 1 rows, one too short) beside long ones: several super-chunks.  With folded column sums (fold 2) the sums of the
    left frames come from the carried items per pack step: the same sums in another fp64 order."""
    import ctypes as C
    import torch
    from msmbuilder_amd import tICA, _lib
    g = torch.Generator(device="cuda").manual_seed(F + lag)
    lens = [9000, 3 * 32 + lag + 1, 40 * 32 + lag + 12, 7000 + lag + 17, 1500 * 2 + lag + 31 - 8, lag + 1, lag, 4096 + lag, 12000, 5000 + lag + 1]
    base = 3.0 * torch.randn(F, generator=g, device="cuda")
    seqs = [(base + torch.randn(n, F, generator=g, device="cuda").cumsum(0) * 0.05
             + torch.randn(n, F, generator=g, device="cuda")) for n in lens]
    if stored == "bfloat16":
        seqs = [s.to(torch.bfloat16) for s in seqs]
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", mode)
    monkeypatch.setenv("MSM_TICA_FOLD", fold)
    monkeypatch.setenv("MSM_TICA_IMG_FUSED", "0")
    out = {}
    for carry in ("1", "2"):      # 2: the same super-chunks and ring halves, every one packed by the pre-pass kernel
        monkeypatch.setenv("MSM_TICA_IMG_CARRY", carry)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = tICA(n_components=3, lag_time=lag).fit(seqs)
            flag = C.c_int(-1)
            _lib.check(_lib.lib().msm_tica_last_img_carried(m._handle, C.byref(flag)))
            assert flag.value == (1 if carry == "1" else 0)
            m.partial_fit(seqs[0])     # on top of existing slabs; a single trajectory: still several super-chunks
        m._pull()
        out[carry] = [np.array(getattr(m, a)) for a in ("_outer_0_to_T_lagged", "_outer_gram_sum", "_sum_0_to_TminusTau",
                                                         "_sum_tau_to_T")] + [np.asarray(m.eigenvalues_)]
        assert m.n_observations_ == sum(n for n in lens if n > lag) + lens[0]
        assert np.abs(out[carry][0]).max() > 0 and np.isfinite(out[carry][1]).all()
    for i, (a, b) in enumerate(zip(out["1"], out["2"])):
        if fold == "0":
            assert np.array_equal(a, b), i
        else:
            np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-11 * np.abs(b).max())


@pytest.mark.parametrize("stored", ["bfloat16", "float32"])
def test_config5_carried_pack_on_trajectory_segments(gpu, monkeypatch, stored):
    """The carried pack under `partial_fit_segments` (one long trajectory cut over ranks; a piece = its owned left
    frames + a halo of `lag` rows): the chunk table holds virtual trajectory bases and slice ends there.  Same accumulators,
    bit for bit, as the same super-chunks packed by the pre-pass kernel; and the pieces add up to the unsplit fit."""
    import ctypes as C
    import torch
    from msmbuilder_amd import tICA, _lib
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "bf16")
    monkeypatch.setenv("MSM_TICA_IMG_FUSED", "0")
    F, lag, n = 1024, 37, 52_000
    g = torch.Generator(device="cuda").manual_seed(9)
    X = (1.5 + torch.randn(n, F, generator=g, device="cuda").cumsum(0) * 0.02 + torch.randn(n, F, generator=g, device="cuda"))
    X = X.to(torch.bfloat16 if stored == "bfloat16" else torch.float32)
    cuts = [0, 20_011, 20_012 + lag, n]
    pieces = [(X[b:min(e + lag, n)], n, b, b, e) for b, e in zip(cuts[:-1], cuts[1:])]
    out = {}
    for carry in ("1", "2"):
        monkeypatch.setenv("MSM_TICA_IMG_CARRY", carry)
        m = tICA(n_components=3, lag_time=lag)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m.partial_fit_segments(pieces)
        flag = C.c_int(-1)
        _lib.check(_lib.lib().msm_tica_last_img_carried(m._handle, C.byref(flag)))
        assert flag.value == (1 if carry == "1" else 0)
        assert m.n_observations_ == n and m.n_sequences_ == 1
        m._pull()
        out[carry] = [np.array(getattr(m, a)) for a in ("_outer_0_to_T_lagged", "_outer_gram_sum", "_sum_0_to_TminusTau", "_sum_tau_to_T")]
    for a, b in zip(out["1"], out["2"]):
        assert np.array_equal(a, b)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        whole = tICA(n_components=3, lag_time=lag).fit([X])
    whole._pull()
    scale = np.abs(whole._outer_gram_sum).max()
    np.testing.assert_allclose(out["1"][1], whole._outer_gram_sum, rtol=0, atol=2e-3 * scale)
    np.testing.assert_allclose(out["1"][2], whole._sum_0_to_TminusTau, rtol=1e-9)


def test_config5_fused_is_the_default_up_to_512_features(gpu, monkeypatch):
    """Round 6 (VERDICT r5 #3a): the fused kernel wins up to 512 features and loses from 768 (profiles/r06_fused_probe.txt), so
    that is where the default switches; MSM_TICA_IMG_FUSED still forces either."""
    import ctypes as C
    import torch
    from msmbuilder_amd import tICA, _lib
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "bf16")
    monkeypatch.delenv("MSM_TICA_IMG_FUSED", raising=False)
    g = torch.Generator(device="cuda").manual_seed(3)
    for F, want in ((256, 1), (512, 1), (768, 0), (1024, 0)):
        X = (torch.randn(6000, F, generator=g, device="cuda") + 1.0).to(torch.bfloat16)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = tICA(n_components=3, lag_time=10).fit([X[:4000], X[4000:]])
        flag = C.c_int(-1)
        _lib.check(_lib.lib().msm_tica_last_img_fused(m._handle, C.byref(flag)))
        assert flag.value == want, (F, flag.value)
        assert np.isfinite(m.eigenvalues_).all()


def test_config5_per_gpu_share_6250000_x_2048_bf16_stored(gpu, monkeypatch):
    """BASELINE configs[4] at the size ONE of its 8 GPUs holds: 6,250,000 x 2048, bfloat16-STORED (25.6 GB), lag 100,
    through the bf16 image path in both of its modes.  Reference: an independent float64 contraction of the stored values
    (torch matmuls on the device, per block of trajectories: different kernels, different summation order), finalised
    with the oracle's formulas and solved on the host.  Tolerances are the modes' stated ones (bf16x2: fp32 class,
    bf16: 8-bit significands)."""
    import torch
    import scipy.linalg
    import bench
    from msmbuilder_amd import tICA
    from oracle.tica_oracle import rao_blackwell_ledoit_wolf
    n_seq, T, F, lag, k = 625, 10_000, 2048, 100, 10
    X = bench.synth_bf16(torch, n_seq, T, F, 11, torch.device("cuda"))
    assert X.dtype == torch.bfloat16 and X.shape == (n_seq * T, F) and X.element_size() * X.numel() == 25_600_000_000
    seqs = list(X.view(n_seq, T, F).unbind(0))
    C = torch.zeros(F, F, dtype=torch.float64, device="cuda")
    G = torch.zeros_like(C)
    s0 = torch.zeros(F, dtype=torch.float64, device="cuda")
    st = torch.zeros_like(s0)
    for s in range(0, n_seq, 25):
        V = X.view(n_seq, T, F)[s:s + 25].double()
        A, B = V[:, :-lag].reshape(-1, F), V[:, lag:].reshape(-1, F)
        C += A.T @ B
        G += A.T @ A + B.T @ B
        s0 += A.sum(0)
        st += B.sum(0)
        del V, A, B
    C, G, s0, st = C.cpu().numpy(), G.cpu().numpy(), s0.cpu().numpy(), st.cpu().numpy()
    two_n = 2.0 * n_seq * (T - lag)
    mu = (s0 + st) / two_n
    S = G / two_n - np.outer(mu, mu)
    OC = (C + C.T) / two_n - np.outer(mu, mu)
    Sig, rho = rao_blackwell_ledoit_wolf(S, n_seq * T)
    ref = scipy.linalg.eigh(OC, b=Sig, subset_by_index=[F - k, F - 1])[0][::-1]
    for mode, rtol, ctol in (("bf16x2", 1e-5, 1e-5), ("bf16", 1e-3, 5e-3)):
        monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", mode)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = tICA(n_components=k, lag_time=lag).fit(seqs)
            ev = np.asarray(m.eigenvalues_)
        assert m._lagged_symmetrised and (m.n_observations_, m.n_sequences_) == (n_seq * T, n_seq)
        np.testing.assert_allclose(ev, ref, rtol=rtol)
        np.testing.assert_allclose(m.means_, mu, rtol=1e-10)
        np.testing.assert_allclose(m.covariance_, Sig, rtol=0, atol=ctol * np.abs(Sig).max())
        np.testing.assert_allclose(m.offset_correlation_, OC, rtol=0, atol=ctol * np.abs(OC).max())
        m._pull()
        np.testing.assert_allclose(m._sum_0_to_TminusTau, s0, rtol=1e-11)
        np.testing.assert_allclose(m._sum_tau_to_T, st, rtol=1e-11)
        del m
        torch.cuda.empty_cache()


@pytest.mark.parametrize("mode,rtol", [("f32", 1e-5), ("f64", 1e-10)])
def test_config3_tica_half_28_x_10000_x_171_vs_oracle(gpu, monkeypatch, mode, rtol):
    """BASELINE configs[2] (Fs peptide: 28 trajectories, contact features) at its full size on the tICA side: 28 x 10,000
    x 171 float32 (171 = the 21-residue contact count, NOT a multiple of 4: the C/G kernel's unaligned-row path),
    tICA(n_components=10, lag_time=1 -- the constructor default the config uses) against the float64 ORACLE, then the
    projection that feeds KCenters(k=200) (whose full-size half is tests/test_gpu_fullsize.py::test_config3_kcenters_280k_bit_exact)."""
    rs = np.random.RandomState(171)
    F, k = 171, 10
    M = rs.randn(6, F) / np.sqrt(6)
    a = np.exp(-1.0 / np.array([400.0, 150.0, 60.0, 25.0, 10.0, 4.0]))
    offs = np.abs(rs.randn(F)) * 0.3 + 0.5                              # contact distances: positive, un-centred
    seqs = []
    for _ in range(28):
        e = rs.randn(10_000, 6)
        z = np.empty_like(e)
        z[0] = e[0]
        for t in range(1, len(e)):
            z[t] = a * z[t - 1] + np.sqrt(1 - a * a) * e[t]
        seqs.append((0.1 * (z @ M) + 0.05 * rs.randn(10_000, F) + offs).astype(np.float32))
    m, o = _fit(seqs, mode, monkeypatch, n_components=k, lag_time=1)
    assert (m.n_observations_, m.n_sequences_, m.n_features) == (280_000, 28, 171)
    # the six slow processes at the stated relative tolerance; the four eigenvalues taken from the noise bulk (0.03, gaps of
    # 1e-3) at the same ABSOLUTE level: the accumulation error is relative to the moments, not to a small eigenvalue
    np.testing.assert_allclose(m.eigenvalues_[:6], o.eigenvalues_[:6], rtol=rtol)
    np.testing.assert_allclose(m.eigenvalues_, o.eigenvalues_, rtol=0, atol=rtol * o.eigenvalues_[0])
    np.testing.assert_allclose(m.means_, o.means_, rtol=1e-10)
    np.testing.assert_allclose(m.covariance_, o.covariance_, rtol=0, atol=rtol * np.abs(o.covariance_).max())
    np.testing.assert_allclose(m.offset_correlation_, o.offset_correlation_, rtol=0, atol=rtol * np.abs(o.offset_correlation_).max())
    Y, Yo = m.transform(seqs[:3]), o.transform(seqs[:3])
    for yh, yo in zip(Y, Yo):
        sign = np.sign((yh * yo).sum(0))
        scale = np.abs(yo).max(0)
        assert yh.shape == (10_000, k)
        assert np.all(np.abs(yh * sign - yo)[:, :6] <= (2e-3 if mode == "f32" else 1e-7) * scale[:6])   # the slow coordinates
