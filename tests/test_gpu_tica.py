"""GPU parity of msmbuilder_amd.tICA (through the C ABI) against the oracle and the
golden vectors produced by the real reference (tests/golden/make_golden.py)."""
import os
import pickle
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = {"f32": 1e-5, "f64": 1e-10, "bf16x2": 1e-5, "bf16": 5e-3}   # stated eigenvalue tolerances; bf16: 1e-3 at >= 1e5 frames (test_config5), 5e-3 on these short ill-conditioned inputs
ATOL_SCALE = {"f32": 1e-6, "bf16x2": 1e-6, "bf16": 5e-4}             # accumulators, relative to max|G| (f32: the sum/difference
# kernel keeps H = G + (C + C^T), up to twice |G|, in fp32 partials of <= 4096 frames; the error is per partial, so relative to the
# totals it shrinks with the number of partials -- these short inputs are the worst case)


def _ar1(seed, n_seq, n_frames, n_features, offset=3.0):
    rs = np.random.RandomState(seed)
    k = 6
    M = rs.randn(k, n_features)
    b = rs.uniform(-offset, offset, size=n_features)
    a = np.exp(-1.0 / (5.0 * (1 + np.arange(k))))
    out = []
    for _ in range(n_seq):
        eps = rs.randn(n_frames, k)
        z = np.zeros((n_frames, k))
        z[0] = eps[0]
        for t in range(1, n_frames):
            z[t] = a * z[t - 1] + np.sqrt(1 - a * a) * eps[t]
        out.append((z.dot(M) + 0.5 * rs.randn(n_frames, n_features) + b).astype(np.float32))
    return out


def _fit_pair(seqs, mode, monkeypatch, **kw):
    from msmbuilder_amd import tICA
    from oracle.tica_oracle import TicaOracle
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", mode)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = tICA(**kw).fit(seqs)
        o = TicaOracle(**kw).fit(seqs)
    return m, o


def _lagged(m, C):
    """What ``_outer_0_to_T_lagged`` must equal: the raw lagged moment, or its symmetric part when the handle runs
    the fp32 sum/difference kernel (msm_tica_lagged_symmetrised)."""
    return 0.5 * (C + C.T) if m._lagged_symmetrised else C


def _vec_match(V, Vref, Sigma, tol=1e-4):
    """eigenvectors up to sign: |v^T Sigma v_ref| >= 1 - tol (both Sigma-orthonormal)."""
    ov = np.abs(np.einsum("ik,ij,jk->k", V, Sigma, Vref))
    assert np.all(ov >= 1 - tol), ov


@pytest.mark.parametrize("mode", ["f32", "f64", "bf16x2", "bf16"])
@pytest.mark.parametrize("F,lag,nf", [(12, 7, 400), (128, 10, 1500), (171, 1, 900), (260, 25, 700)])
def test_accumulators_and_eigs_vs_oracle(gpu, monkeypatch, mode, F, lag, nf):
    seqs = _ar1(F + lag, 4, nf, F)
    seqs[1] = seqs[1][: nf // 3]        # ragged
    seqs.append(seqs[0][:lag])          # == lag: skipped (tica.py:410-412)
    seqs.append(seqs[2][: lag + 1])     # one lagged pair
    m, o = _fit_pair(seqs, mode, monkeypatch, n_components=5, lag_time=lag)
    assert m.n_observations_ == o.n_observations_ and m.n_sequences_ == o.n_sequences_
    m._pull()
    # fp32 chunk partials: |err| <= ~1e-7 * sum|a*b|, i.e. relative to the matrix SCALE, not per element
    G = o.S0 + o.Stau
    tol = dict(rtol=1e-12, atol=1e-9) if mode == "f64" else dict(rtol=0, atol=ATOL_SCALE[mode] * np.abs(G).max())
    np.testing.assert_allclose(m._outer_0_to_T_lagged, _lagged(m, o.C), **tol)
    np.testing.assert_allclose(m._outer_gram_sum, G, **tol)
    np.testing.assert_allclose(m._sum_0_to_TminusTau, o.s0, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(m._sum_tau_to_T, o.stau, rtol=1e-12, atol=1e-9)
    assert np.array_equal(m._outer_gram_sum, m._outer_gram_sum.T)
    np.testing.assert_allclose(m.eigenvalues_, o.eigenvalues_, rtol=RTOL[mode])
    np.testing.assert_allclose(m.means_, o.means_, rtol=1e-10, atol=1e-12)
    assert abs(m.shrinkage_ - o.shrinkage_) <= (1e-3 if mode == "bf16" else 1e-6) * max(1e-12, abs(o.shrinkage_)) + 1e-12
    if mode != "bf16":   # bf16 input rounding rotates eigenvectors inside near-degenerate clusters of eigenvalues
        _vec_match(m.eigenvectors_, o.eigenvectors_, o.covariance_)


@pytest.mark.parametrize("mode", ["f32", "f64"])
def test_golden_known_answers(gpu, monkeypatch, golden_dir, mode):
    """Outputs of the REAL reference tica.py (fixtures) incl. SURVEY.md's known answers."""
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", mode)
    g = np.load(os.path.join(golden_dir, "tica_golden.npz"))
    rs = np.random.RandomState(0)
    seqs = [rs.randn(1000, 6).astype(np.float32) for _ in range(3)] + [rs.randn(2, 6).astype(np.float32)]
    for tag, shr in (("A0", 0), ("An", None)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = tICA(n_components=3, lag_time=2, shrinkage=shr).fit(seqs)
        assert [m.n_observations_, m.n_sequences_] == list(g[tag + "_n_obs_seq"])
        np.testing.assert_allclose(m.eigenvalues_, g[tag + "_eigenvalues"], rtol=RTOL[mode] * 10)
        np.testing.assert_allclose(m.timescales_, g[tag + "_timescales"], rtol=1e-4)
        np.testing.assert_allclose(m.means_, g[tag + "_means"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(m.offset_correlation_, g[tag + "_offset_correlation"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(m.covariance_, g[tag + "_covariance"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(m.shrinkage_, g[tag + "_shrinkage_"], rtol=1e-5, atol=1e-12)
        np.testing.assert_allclose(m.score_, g[tag + "_score_"], rtol=1e-5)
        Y = m.transform(seqs[:1])[0]
        Yg = g[tag + "_transform0"]
        assert Y.dtype == np.float64 and Y.shape == Yg.shape
        sign = np.sign((Y * Yg).sum(0))
        np.testing.assert_allclose(Y * sign, Yg, rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(m.score(seqs[1:3]), g[tag + "_score_test"], rtol=1e-4)
    # SURVEY.md section 8(c) constants
    np.testing.assert_allclose(g["A0_eigenvalues"], [0.030889057382, 0.024348710243, 0.010004331246], rtol=1e-9)


def test_golden_ragged_mappings(gpu, monkeypatch, golden_dir):
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f64")
    g = np.load(os.path.join(golden_dir, "tica_golden.npz"))
    seqs = [g["B_seq%d" % i] for i in range(5)]
    for tag, kw in (("B", {}), ("Bk", dict(kinetic_mapping=True)), ("Bc", dict(commute_mapping=True))):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = tICA(n_components=4, lag_time=7, **kw).fit(seqs)
        np.testing.assert_allclose(m.eigenvalues_, g[tag + "_eigenvalues"], rtol=1e-9)
        np.testing.assert_allclose(m.timescales_, g[tag + "_timescales"], rtol=1e-7)
        Y, Yg = m.transform(seqs[1:2])[0], g[tag + "_transform1"]
        sign = np.sign((Y * Yg).sum(0))
        sign[sign == 0] = 1
        np.testing.assert_allclose(Y * sign, Yg, rtol=1e-6, atol=1e-8)
        if tag == "B":
            m._pull()
            np.testing.assert_allclose(m._outer_0_to_T_lagged, _lagged(m, g["B_C"]), rtol=1e-12, atol=1e-9)
            np.testing.assert_allclose(m._outer_gram_sum, g["B_S0"] + g["B_Stau"], rtol=1e-12, atol=1e-9)
            np.testing.assert_allclose(m._sum_0_to_TminusTau, g["B_s0"], rtol=1e-12, atol=1e-9)
            assert [m.n_observations_, m.n_sequences_] == list(g["B_n_obs_seq"])
            mine, ref = m.summarize().splitlines(), str(g["B_summarize"]).splitlines()
            assert mine[:3] == ref[:3] and mine[4:7] == ref[4:7]     # line 3 is shrinkage (float repr, 1 ulp)
            assert mine[3].split(":")[0] == ref[3].split(":")[0]


def test_reference_identities(gpu, monkeypatch):
    """Restated from the reference's own tests (tests/test_decomposition.py:59-110,
    tests/test_utils.py:59-79)."""
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f64")
    rs = np.random.RandomState(42)
    X = rs.randn(100, 5)
    for n in range(1, 5):
        t = tICA(n_components=n, shrinkage=0).fit([X])
        np.testing.assert_approx_equal(t.score([X]), t.eigenvalues_.sum())
        np.testing.assert_approx_equal(t.score([X]), t.score_)
    # changing n_components after the fit
    t = tICA(n_components=1, shrinkage=0).fit([X])
    Y1 = t.transform([X])[0]
    t.n_components = 4
    Y4 = t.transform([X])[0]
    t.n_components = 3
    Y3 = t.transform([X])[0]
    assert Y1.shape == (100, 1) and Y4.shape == (100, 4) and Y3.shape == (100, 3)
    np.testing.assert_allclose(Y1.flatten(), Y3[:, 0], rtol=1e-9)
    np.testing.assert_allclose(Y3, Y4[:, :3], rtol=1e-9)
    # kinetic mapping
    X = rs.randn(10, 3)
    y1 = tICA(n_components=2, lag_time=1).fit_transform([np.copy(X)])[0]
    t2 = tICA(n_components=2, lag_time=1, kinetic_mapping=True)
    y2 = t2.fit_transform([np.copy(X)])[0]
    np.testing.assert_allclose(y2, y1 * t2.eigenvalues_, rtol=1e-9)
    # singular input keeps float64 outputs (test_decomposition.py:28-49)
    Xs = rs.randn(100, 2)
    Xs = np.hstack((Xs, Xs[:, 0, np.newaxis]))
    t = tICA(n_components=1).fit([Xs])
    assert t.components_.dtype == np.float64 and t.eigenvalues_.dtype == np.float64
    # shapes
    model = tICA(n_components=3).fit([rs.randn(100, 10)])
    assert model.eigenvalues_.shape == (3,) and model.eigenvectors_.shape == (10, 3)
    assert model.components_.shape == (3, 10)


def test_errors_and_warnings(gpu):
    from msmbuilder_amd import tICA
    with pytest.raises(ValueError):
        tICA(kinetic_mapping=True, commute_mapping=True)
    with pytest.raises(ValueError):
        tICA().fit([np.zeros(5)])  # not 2-D
    with pytest.raises(ValueError, match="shorter"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tICA(lag_time=10).fit([np.random.randn(10, 3), np.random.randn(4, 3)])
    with pytest.warns(UserWarning, match="too short"):
        tICA(lag_time=10).partial_fit(np.random.randn(5, 3))
    with pytest.raises(RuntimeError):
        tICA().eigenvalues_
    X = np.random.randn(50, 3).astype(np.float32)
    X[7, 1] = np.nan
    t = tICA(lag_time=2).fit([np.random.randn(50, 3).astype(np.float32)])
    before = t.eigenvalues_.copy()
    with pytest.raises(ValueError, match="NaN"):
        t.partial_fit(X)
    t._is_dirty = True
    np.testing.assert_array_equal(t.eigenvalues_, before)   # state untouched by the rejected input


def test_partial_fit_pickle_and_order(gpu, monkeypatch):
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f64")
    seqs = _ar1(5, 6, 300, 20)
    a = tICA(n_components=3, lag_time=4).fit(seqs)
    b = tICA(n_components=3, lag_time=4)
    for s in seqs[:3]:
        b.partial_fit(s)
    b = pickle.loads(pickle.dumps(b))     # resumable after unpickling (plain-pickle models)
    for s in seqs[3:]:
        b.partial_fit(s)
    np.testing.assert_allclose(a.eigenvalues_, b.eigenvalues_, rtol=1e-11)
    assert a.n_observations_ == b.n_observations_ and a.n_sequences_ == b.n_sequences_
    c = pickle.loads(pickle.dumps(a))
    np.testing.assert_array_equal(a.eigenvalues_, c.eigenvalues_)


def test_device_resident_inputs(gpu, monkeypatch):
    torch = pytest.importorskip("torch")
    from msmbuilder_amd import tICA
    seqs = _ar1(9, 3, 500, 64)
    host = tICA(n_components=4, lag_time=3).fit(seqs)
    dev = tICA(n_components=4, lag_time=3).fit([torch.from_numpy(s).cuda() for s in seqs])
    np.testing.assert_array_equal(host.eigenvalues_, dev.eigenvalues_)
    Yd = dev.transform([torch.from_numpy(seqs[0]).cuda()])[0]
    assert Yd.is_cuda and Yd.dtype == torch.float64
    np.testing.assert_array_equal(Yd.cpu().numpy(), host.transform(seqs[:1])[0])


def test_f64_mfma_layout_asymmetric(gpu, monkeypatch):
    """Transpose-detecting check of both MFMA output maps: C must equal X0^T X1, not its transpose."""
    from msmbuilder_amd import tICA
    rs = np.random.RandomState(3)
    X = rs.randn(257, 130).astype(np.float32)
    for mode in ("f32", "f64", "bf16x2", "bf16"):
        monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", mode)
        m = tICA(lag_time=3).fit([X])
        m._pull()
        Xd = X.astype(np.float64)
        C = Xd[:-3].T @ Xd[3:]
        assert np.abs(C - C.T).max() > 1.0
        np.testing.assert_allclose(m._outer_0_to_T_lagged, _lagged(m, C), rtol=1e-5, atol=1e-3 if mode != "bf16" else 0.5)


@pytest.mark.parametrize("F,lag", [(256, 1), (260, 37), (512, 100), (516, 5), (1024, 250), (1284, 3), (2048, 20)])
def test_symmetric_sum_difference_kernel(gpu, monkeypatch, F, lag):
    """fp32 default from 2 tiles (F > 128) on: H = sum u u^T, D = sum d d^T of the upper tiles, G = (H + D)/2,
    (C + C^T)/2 = (H - D)/4.  Against float64 numpy, and against the C/G kernel (MSM_TICA_SYM=0), with trajectories
    that cross the 4096-frame chunks, end inside a 32-frame step, are not longer than the lag, or hold one pair."""
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f32")
    lens = [9001, 4096 + lag, lag, lag + 1, 33 + lag, 1, 2 * lag + 5]
    seqs = [x[:n] for x, n in zip(_ar1(F + lag, len(lens), max(lens), F), lens)]
    m = tICA(n_components=4, lag_time=lag).fit(seqs)
    assert m._lagged_symmetrised
    m._pull()
    monkeypatch.setenv("MSM_TICA_SYM", "0")
    m0 = tICA(n_components=4, lag_time=lag).fit(seqs)
    assert not m0._lagged_symmetrised
    m0._pull()
    C = np.zeros((F, F)); G = np.zeros((F, F)); s0 = np.zeros(F); st = np.zeros(F); n = 0
    for x in seqs:
        if len(x) <= lag:
            continue
        x = x.astype(np.float64)
        a, b = x[:-lag], x[lag:]
        C += a.T @ b; G += a.T @ a + b.T @ b; s0 += a.sum(0); st += b.sum(0); n += len(x)      # n_observations_ counts frames (tica.py:414)
    assert m.n_observations_ == n == m0.n_observations_
    scale = np.abs(G).max()
    np.testing.assert_allclose(m._outer_gram_sum, G, rtol=0, atol=ATOL_SCALE["f32"] * scale)
    np.testing.assert_allclose(m._outer_0_to_T_lagged, 0.5 * (C + C.T), rtol=0, atol=ATOL_SCALE["f32"] * scale)
    np.testing.assert_allclose(m0._outer_0_to_T_lagged, C, rtol=0, atol=ATOL_SCALE["f32"] * scale)
    np.testing.assert_allclose(m._sum_0_to_TminusTau, s0, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(m._sum_tau_to_T, st, rtol=1e-12, atol=1e-9)
    assert np.array_equal(m._outer_gram_sum, m._outer_gram_sum.T)
    assert np.array_equal(m._outer_0_to_T_lagged, m._outer_0_to_T_lagged.T)
    # at lags far beyond the slowest mode, or with only ~6 frames per feature, the leading eigenvalues are a cluster of
    # sampling noise on an ill-conditioned covariance: two fp32 roundings agree absolutely, not to 1e-5 relative
    loose = lag >= 50 or F >= 2048
    np.testing.assert_allclose(m.eigenvalues_, m0.eigenvalues_, rtol=0 if loose else RTOL["f32"], atol=5e-5 if loose else 0)
    np.testing.assert_allclose(m.offset_correlation_, m0.offset_correlation_, rtol=0,
                               atol=2 * ATOL_SCALE["f32"] * scale / n)
    # float64 input to the same handle takes the fp64 kernel (raw C): the export is then the sum of both parts and
    # its symmetric part is still exact
    m.partial_fit(seqs[0].astype(np.float64))
    m._pull()
    x = seqs[0].astype(np.float64)
    C2 = C + x[:-lag].T @ x[lag:]
    got = m._outer_0_to_T_lagged
    np.testing.assert_allclose(0.5 * (got + got.T), 0.5 * (C2 + C2.T), rtol=0, atol=ATOL_SCALE["f32"] * scale)


# ---------------------------------------------------------------- trajectory segments
@pytest.mark.parametrize("mode,rtol", [("f64", 1e-11), ("f32", 2e-6)])
def test_segments_of_one_trajectory_sum_to_the_whole(gpu, monkeypatch, mode, rtol):
    """Cutting one trajectory into pieces owned by different models (= ranks) and summing the
    exported accumulators reproduces the unsplit partial_fit (tica.py:417-422)."""
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", mode)
    rs = np.random.RandomState(5)
    n, F, lag = 9137, 24, 13
    X = (rs.randn(n, F) * 0.8 + 0.2).astype(np.float32)
    whole = tICA(n_components=4, lag_time=lag).fit([X])
    cuts = [0, 1, 2999, 3000 + lag - 1, 9130, n]       # includes a 1-row piece and a piece inside the last lag rows
    parts = []
    for b, e in zip(cuts[:-1], cuts[1:]):
        end = min(e + lag, n)
        m = tICA(n_components=4, lag_time=lag)
        m.partial_fit_segments([(X[b:end], n, b, b, e)])
        parts.append(m)
    assert sum(m.n_observations_ for m in parts) == n and sum(m.n_sequences_ for m in parts) == 1
    for m in parts:
        m._pull()
    whole._pull()
    scale = np.abs(whole._outer_gram_sum).max()
    for name in ("_outer_0_to_T_lagged", "_outer_gram_sum", "_sum_0_to_TminusTau", "_sum_tau_to_T"):
        got = sum(getattr(m, name) for m in parts)
        np.testing.assert_allclose(got, getattr(whole, name), rtol=rtol, atol=rtol * scale, err_msg=name)
    # the same pieces into ONE model, device-resident slices
    import torch
    Xd = torch.from_numpy(X).cuda()
    one = tICA(n_components=4, lag_time=lag)
    one.partial_fit_segments([(Xd[b:min(e + lag, n)], n, b, b, e) for b, e in zip(cuts[:-1], cuts[1:])])
    assert one.n_observations_ == n and one.n_sequences_ == 1
    np.testing.assert_allclose(one.eigenvalues_, whole.eigenvalues_, rtol=max(rtol, 1e-10) * 10)


def test_segment_without_its_halo_is_rejected(gpu):
    from msmbuilder_amd import tICA
    X = np.random.RandomState(0).randn(500, 8).astype(np.float32)
    m = tICA(lag_time=10)
    with pytest.raises(ValueError, match="need rows"):
        m.partial_fit_segments([(X[100:200], 500, 100, 100, 200)])    # needs rows up to 209
    with pytest.raises(ValueError, match="need rows"):
        m.partial_fit_segments([(X[100:220], 500, 100, 90, 200)])     # owns rows it does not hold
    m.partial_fit_segments([(X[100:210], 500, 100, 100, 200)])
    assert m.n_observations_ == 100 and m.n_sequences_ == 0


# ---------------------------------------------------------------- device eigensolve
@pytest.mark.parametrize("F,k", [(7, 7), (64, 5), (300, 10), (1100, 12)])
def test_device_eigensolve_matches_lapack(gpu, F, k):
    import scipy.linalg
    from msmbuilder_amd.decomposition import _moments
    rs = np.random.RandomState(F)
    A = rs.randn(F, 3 * F)
    S = A.dot(A.T) / (3 * F) + 0.05 * np.eye(F)
    B = rs.randn(F, F)
    OC = 0.4 * S + 0.03 * (B + B.T)
    vals, vecs = _moments.device_generalized_eigenpairs(OC, S, k)
    w, v = scipy.linalg.eigh(OC, b=S, subset_by_index=[F - k, F - 1])
    np.testing.assert_allclose(vals, w[::-1], rtol=1e-10, atol=1e-12)
    v = v[:, ::-1]
    np.testing.assert_allclose(np.abs(np.sum(vecs * S.dot(v), axis=0)), 1.0, rtol=1e-8)   # same vectors up to sign, B-orthonormal
    np.testing.assert_allclose(vecs.T.dot(S).dot(vecs), np.eye(k), atol=1e-9)
    with pytest.raises(np.linalg.LinAlgError):
        _moments.device_generalized_eigenpairs(OC, S - 10 * np.eye(F), k)


def test_tica_with_device_solve(gpu, monkeypatch):
    from msmbuilder_amd import tICA
    seqs = _ar1(3, 3, 1500, 16)
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f64")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        monkeypatch.setenv("MSMBUILDER_AMD_DEVICE_SOLVE", "0")
        host = tICA(n_components=4, lag_time=3).fit(seqs)
        ev_h, y_h = host.eigenvalues_, host.transform(seqs[:1])[0]
        monkeypatch.setenv("MSMBUILDER_AMD_DEVICE_SOLVE", "1")
        dev = tICA(n_components=4, lag_time=3).fit(seqs)
        ev_d, y_d = dev.eigenvalues_, dev.transform(seqs[:1])[0]
    np.testing.assert_allclose(ev_d, ev_h, rtol=1e-11)
    s = np.sign(np.sum(y_d * y_h, axis=0))
    np.testing.assert_allclose(y_d * s, y_h, rtol=1e-7, atol=1e-9)


# ---------------------------------------------------------------- projection kernels (tica.py:329-352)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,F,k", [(1, 4, 1), (255, 20, 3), (257, 36, 10), (1000, 100, 16), (513, 512, 17), (300, 130, 40), (64, 6, 6)])
def test_project_matches_numpy(gpu, dtype, n, F, k):
    """msm_tica_project (fp64-MFMA kernel for 16-byte aligned rows, lane-per-row kernel otherwise)
    against (X - mu) . V^T in float64."""
    import ctypes as C
    import torch
    from msmbuilder_amd import _lib
    from msmbuilder_amd._lib import Arr, check
    rs = np.random.RandomState(n + F + k)
    X = (rs.randn(n, F) * 3 + 1).astype(dtype)
    mu = rs.randn(F)
    V = rs.randn(k, F)
    want = (X.astype(np.float64) - mu).dot(V.T)
    for dev in (False, True):
        src = torch.from_numpy(X).cuda() if dev else X
        ax = Arr(src)
        out = _lib.empty_like_placement(ax, (n, k), np.float64)
        ao = Arr(out, np.float64)
        check(_lib.lib().msm_tica_project(ax.vp, ax.dtype.itemsize, n, F, F, mu.ctypes.data, np.ascontiguousarray(V).ctypes.data,
                                          k, ao.vp, ax.on_device, 1))
        got = out.cpu().numpy() if dev else out
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-11 * np.abs(want).max())
    Xb = X.copy()
    Xb[n // 2, F - 1] = np.inf
    ax = Arr(Xb)
    out = np.zeros((n, k))
    rc = _lib.lib().msm_tica_project(ax.vp, ax.dtype.itemsize, n, F, F, mu.ctypes.data, np.ascontiguousarray(V).ctypes.data, k,
                                     Arr(out, np.float64).vp, 0, 1)
    assert rc == _lib.MSM_ERR_NONFINITE


@pytest.mark.parametrize("n,F,k", [(1, 8, 1), (255, 24, 3), (1000, 104, 16), (513, 512, 17), (300, 136, 40), (700, 2048, 10),
                                   (257, 36, 10), (64, 6, 6), (129, 71, 5)])
def test_project_bfloat16_stored_rows(gpu, n, F, k):
    """bfloat16-STORED rows (BASELINE configs[4]) are widened INSIDE the projection kernels (dtype_bytes = 2; 64-feature
    chunks on the fp64 matrix pipe for rows of a multiple of 8 features, the lane-per-row kernel otherwise): the widening
    is exact, so the result equals (X - mu) . V^T of the stored values in float64, and `transform` of a bfloat16 tensor
    equals `transform` of its float32 image bit for bit.  A non-finite stored value is reported like any other."""
    import ctypes as C
    import torch
    from msmbuilder_amd import _lib, tICA
    rs = np.random.RandomState(n + F + k)
    Xb = torch.from_numpy((rs.randn(n, F) * 3 + 1).astype(np.float32)).cuda().to(torch.bfloat16)
    X32 = Xb.float()
    mu = rs.randn(F)
    V = np.ascontiguousarray(rs.randn(k, F))
    want = (X32.cpu().numpy().astype(np.float64) - mu).dot(V.T)
    L = _lib.lib()
    _lib.ensure_device(0)
    _lib.set_stream(torch.cuda.current_stream().cuda_stream)

    def project(t, nbytes):
        out = torch.empty((n, k), dtype=torch.float64, device="cuda")
        rc = L.msm_tica_project(C.c_void_p(t.data_ptr()), nbytes, n, F, F, mu.ctypes.data, V.ctypes.data, k,
                                C.c_void_p(out.data_ptr()), 1, 1)
        return rc, out
    rc, got = project(Xb, 2)
    assert rc == 0
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-12, atol=1e-11 * np.abs(want).max())
    rc32, got32 = project(X32, 4)
    assert rc32 == 0
    if F % 8 == 0 or F % 4 != 0:   # the same kernel family for both element types (both MFMA, or both lane-per-row)
        np.testing.assert_allclose(got.cpu().numpy(), got32.cpu().numpy(), rtol=1e-13, atol=1e-12 * np.abs(want).max())
    bad = Xb.clone()
    bad[n // 2, F - 1] = float("nan")
    rc, _ = project(bad, 2)
    assert rc == _lib.MSM_ERR_NONFINITE
    if n > 64 and F >= 24:
        m = tICA(n_components=min(3, F), lag_time=2)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m.fit([X32])
            ya, yb = m.transform([Xb])[0], m.transform([X32])[0]
        assert ya.dtype == torch.float64 and ya.is_cuda
        if F % 8 == 0:
            np.testing.assert_allclose(ya.cpu().numpy(), yb.cpu().numpy(), rtol=1e-12, atol=1e-12 * float(yb.abs().max()))


@pytest.mark.parametrize("F", [132, 256, 388])
def test_symmetric_kernel_half_step_edges(gpu, monkeypatch, F):
    """Pair counts around the kernel's 16-frame half-steps and 32-frame steps, one trajectory each and all together
    (chunks that start and end inside a step), against float64 numpy."""
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f32")
    lag = 3
    rs = np.random.RandomState(F)
    pairs = [1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 4095, 4096, 4097]
    seqs = [(rs.randn(n + lag, F) * 2 + rs.randn(F)).astype(np.float32) for n in pairs]

    def ref(ss):
        C = np.zeros((F, F)); G = np.zeros((F, F))
        for x in ss:
            x = x.astype(np.float64)
            a, b = x[:-lag], x[lag:]
            C += a.T @ b; G += a.T @ a + b.T @ b
        return 0.5 * (C + C.T), G

    for group in [[s] for s in seqs[:14]] + [seqs]:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = tICA(lag_time=lag).fit(group)
        assert m._lagged_symmetrised
        m._pull()
        Cs, G = ref(group)
        scale = max(np.abs(G).max(), 1e-30)
        np.testing.assert_allclose(m._outer_gram_sum, G, rtol=0, atol=ATOL_SCALE["f32"] * scale)
        np.testing.assert_allclose(m._outer_0_to_T_lagged, Cs, rtol=0, atol=ATOL_SCALE["f32"] * scale)


def test_solve_paths_agree(gpu, monkeypatch):
    """host (numpy + dsygvx), hybrid (device finalise + Cholesky reduction + back-substitution; here 6 features: LAPACK's
    dsyevr on the reduced matrix), all-device rocSOLVER: one model."""
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f64")
    seqs = _ar1(5, 6, 3000, 200)
    out = {}
    for name, env in (("host", {"MSMBUILDER_AMD_DEVICE_SOLVE": "0"}),
                      ("hybrid", {"MSMBUILDER_AMD_DEVICE_SOLVE": "hybrid"}),
                      ("dev", {"MSMBUILDER_AMD_DEVICE_SOLVE": "1"})):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        m = tICA(n_components=6, lag_time=9).fit(seqs)
        out[name] = (m.eigenvalues_.copy(), m.eigenvectors_.copy(), m.shrinkage_, m.means_.copy())
    for name in ("hybrid", "dev"):
        np.testing.assert_allclose(out[name][0], out["host"][0], rtol=1e-11)
        sg = np.sign((out[name][1] * out["host"][1]).sum(0))
        np.testing.assert_allclose(out[name][1] * sg, out["host"][1], rtol=0, atol=1e-8 * np.abs(out["host"][1]).max())
        assert abs(out[name][2] - out["host"][2]) <= 1e-12 * abs(out["host"][2])
        np.testing.assert_allclose(out[name][3], out["host"][3], rtol=1e-13)
