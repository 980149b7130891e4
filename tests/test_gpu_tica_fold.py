"""The sum/difference kernel with the column sums folded into its staging lanes (no column-sum pass over X ahead of the
MFMA kernel; csrc/tica.hip, FOLD): same sums (tica.py:418-419), same moments, same finite check with a rejected input
leaving the state untouched (utils/validation.py:68-74 raises before tica.py:401 accumulates).  MSM_TICA_FOLD=2
lets small inputs take the path that by default starts at 2^26 elements."""
import ctypes as C
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _tile_kernel(monkeypatch):
    """The folded column sums belong to the 128-tile sum/difference kernel; at F <= 256 the whole-matrix kernel of round 6
    (tica_symw_dev.h, which has a column-sum pass of its own ahead of it) would take these shapes: switched off here."""
    monkeypatch.setenv("MSM_TICA_SYMW", "0")

ATOL_SCALE = 1e-6   # accumulators relative to max|G| (tests/test_gpu_tica.py)


def _folded(m):
    from msmbuilder_amd import _lib
    f = C.c_int(-1)
    _lib.check(_lib.lib().msm_tica_last_folded(m._handle, C.byref(f)))
    return f.value


def _data(seed, lens, F, offset=3.0):
    rs = np.random.RandomState(seed)
    b = rs.uniform(-offset, offset, size=F)
    M = rs.randn(5, F)
    out = []
    for n in lens:
        z = np.cumsum(rs.randn(n, 5), axis=0) * 0.05 + rs.randn(n, 5)
        out.append((z.dot(M) + 0.5 * rs.randn(n, F) + b).astype(np.float32))
    return out


def _numpy_moments(seqs, lag, F):
    Cm = np.zeros((F, F)); G = np.zeros((F, F)); s0 = np.zeros(F); st = np.zeros(F); n = 0
    for x in seqs:
        if len(x) <= lag:
            continue
        x = x.astype(np.float64)
        a, b = x[:-lag], x[lag:]
        Cm += a.T @ b; G += a.T @ a + b.T @ b; s0 += a.sum(0); st += b.sum(0); n += len(x)
    return Cm, G, s0, st, n


@pytest.mark.parametrize("F,lag", [(256, 1), (256, 37), (512, 100), (1024, 16)])
def test_folded_sums_and_moments_vs_numpy(gpu, monkeypatch, F, lag):
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f32")
    monkeypatch.setenv("MSM_TICA_FOLD", "2")
    # chunks that end inside a step, trajectories of exactly 2 lag frames (every frame is a boundary frame), one too short
    # to count (skipped), half-step edges
    lens = [9001, 4096 + 2 * lag, lag, 2 * lag, 33 + 2 * lag, 2 * lag + 5, 4097, 2 * lag + 16, 2 * lag + 17]
    seqs = _data(F + lag, lens, F)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = tICA(n_components=4, lag_time=lag).fit(seqs)
    assert m._lagged_symmetrised and _folded(m) == 1
    m._pull()
    Cm, G, s0, st, n = _numpy_moments(seqs, lag, F)
    assert m.n_observations_ == n
    scale = np.abs(G).max()
    np.testing.assert_allclose(m._sum_0_to_TminusTau, s0, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(m._sum_tau_to_T, st, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(m._outer_gram_sum, G, rtol=0, atol=ATOL_SCALE * scale)
    np.testing.assert_allclose(m._outer_0_to_T_lagged, 0.5 * (Cm + Cm.T), rtol=0, atol=ATOL_SCALE * scale)
    # and against the same handle type with the separate column-sum pass
    monkeypatch.setenv("MSM_TICA_FOLD", "0")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m0 = tICA(n_components=4, lag_time=lag).fit(seqs)
    assert _folded(m0) == 0
    m0._pull()
    np.testing.assert_allclose(m._sum_0_to_TminusTau, m0._sum_0_to_TminusTau, rtol=1e-13, atol=1e-10)
    np.testing.assert_allclose(m._sum_tau_to_T, m0._sum_tau_to_T, rtol=1e-13, atol=1e-10)
    np.testing.assert_allclose(m.means_, m0.means_, rtol=1e-12, atol=1e-13)
    loose = lag >= 50
    np.testing.assert_allclose(m.eigenvalues_, m0.eigenvalues_, rtol=0 if loose else 1e-5, atol=5e-5 if loose else 0)
    np.testing.assert_allclose(m.covariance_, m0.covariance_, rtol=0, atol=2 * ATOL_SCALE * scale / n)


def test_short_trajectory_falls_back_to_the_column_sum_pass(gpu, monkeypatch):
    """A trajectory of lag < len < 2 lag frames has rows that are neither a left nor a right frame's complement of the
    boundary rows: such a launch keeps the separate pass."""
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f32")
    monkeypatch.setenv("MSM_TICA_FOLD", "2")
    F, lag = 256, 40
    seqs = _data(3, [3000, lag + 7, 500], F)
    m = tICA(n_components=3, lag_time=lag).fit(seqs)
    assert _folded(m) == 0
    m._pull()
    Cm, G, s0, st, n = _numpy_moments(seqs, lag, F)
    np.testing.assert_allclose(m._sum_0_to_TminusTau, s0, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(m._outer_gram_sum, G, rtol=0, atol=ATOL_SCALE * np.abs(G).max())
    # features that do not fill whole tiles: not folded either
    m = tICA(n_components=3, lag_time=5).fit(_data(4, [2000, 900], 260))
    assert _folded(m) == 0


def test_default_threshold(gpu, monkeypatch):
    """Below 2^26 elements a launch keeps the column-sum pass (its cost there is microseconds; results of small fits do not
    depend on a sampled shift row); from there on it folds."""
    torch = pytest.importorskip("torch")
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f32")
    monkeypatch.delenv("MSM_TICA_FOLD", raising=False)
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randn(300000, 256, device="cuda", generator=g)
    m = tICA(n_components=3, lag_time=10).fit([X[:200000]])
    assert _folded(m) == 0
    m.partial_fit(X)        # 76.8M elements
    assert _folded(m) == 1
    m._pull()
    s0 = (X[:200000 - 10].double().sum(0) + X[:-10].double().sum(0)).cpu().numpy()
    st = (X[10:200000].double().sum(0) + X[10:].double().sum(0)).cpu().numpy()
    np.testing.assert_allclose(m._sum_0_to_TminusTau, s0, rtol=1e-11, atol=1e-7)
    np.testing.assert_allclose(m._sum_tau_to_T, st, rtol=1e-11, atol=1e-7)
    a, b = X[:-10].double(), X[10:].double()
    a2, b2 = X[:200000 - 10].double(), X[10:200000].double()
    G = (a.T @ a + b.T @ b + a2.T @ a2 + b2.T @ b2).cpu().numpy()
    np.testing.assert_allclose(m._outer_gram_sum, G, rtol=0, atol=ATOL_SCALE * np.abs(G).max())


@pytest.mark.parametrize("where", ["interior", "first_rows", "last_rows", "inf"])
@pytest.mark.parametrize("dirty", [False, True])
def test_rejected_input_leaves_the_state_untouched(gpu, monkeypatch, where, dirty):
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f32")
    monkeypatch.setenv("MSM_TICA_FOLD", "2")
    F, lag = 256, 9
    good = _data(11, [5000, 700], F)
    bad = _data(12, [4500], F)[0]
    row = {"interior": 2311, "first_rows": 3, "last_rows": 4500 - 2, "inf": 1200}[where]
    bad[row, 77] = np.inf if where == "inf" else np.nan
    m = tICA(n_components=3, lag_time=lag)
    if dirty:
        m.fit(good)
        assert _folded(m) == 1
        m._pull()
        before = [m._outer_gram_sum.copy(), m._outer_0_to_T_lagged.copy(), m._sum_0_to_TminusTau.copy(),
                  m._sum_tau_to_T.copy(), m.n_observations_]
    with pytest.raises(ValueError, match="NaN"):
        m.partial_fit(bad)
    if dirty:
        m._is_dirty = True
        m._pull()
        after = [m._outer_gram_sum, m._outer_0_to_T_lagged, m._sum_0_to_TminusTau, m._sum_tau_to_T, m.n_observations_]
        for x, y in zip(before, after):
            np.testing.assert_array_equal(x, y)
    # the model goes on as if the rejected call had never been made
    fresh = tICA(n_components=3, lag_time=lag)
    if dirty:
        fresh.fit(good)
    more = _data(13, [3000], F)[0]
    m.partial_fit(more)
    fresh.partial_fit(more)
    assert _folded(m) == 1
    m._pull(); fresh._pull()
    np.testing.assert_array_equal(m._outer_gram_sum, fresh._outer_gram_sum)
    np.testing.assert_array_equal(m._sum_tau_to_T, fresh._sum_tau_to_T)
    np.testing.assert_array_equal(m.eigenvalues_, fresh.eigenvalues_)


def test_folded_and_plain_launches_add_up(gpu, monkeypatch):
    """partial_fit calls of both kinds on one handle (the shift row is whatever the first call made it) against one fit."""
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f32")
    F, lag = 384, 12
    seqs = _data(21, [6000, 2500, 4100, 1500], F, offset=50.0)     # |mean| / sigma up to ~20
    monkeypatch.setenv("MSM_TICA_FOLD", "2")
    m = tICA(n_components=4, lag_time=lag, shrinkage=0)
    kinds = []
    for i, s in enumerate(seqs):
        monkeypatch.setenv("MSM_TICA_FOLD", "2" if i % 2 == 0 else "0")
        m.partial_fit(s)
        kinds.append(_folded(m))
    assert kinds == [1, 0, 1, 0]
    m._pull()
    Cm, G, s0, st, n = _numpy_moments(seqs, lag, F)
    scale = np.abs(G).max()
    np.testing.assert_allclose(m._sum_0_to_TminusTau, s0, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(m._sum_tau_to_T, st, rtol=1e-12, atol=1e-9)
    # un-centred features: the raw moments are dominated by n mu mu^T, which the fp64 restoration carries exactly; the
    # covariance is what shows the kernel's rounding
    X0 = np.concatenate([s[:-lag] for s in seqs]).astype(np.float64)
    X1 = np.concatenate([s[lag:] for s in seqs]).astype(np.float64)
    mu = (X0.sum(0) + X1.sum(0)) / (2 * len(X0))
    cov = ((X0 - mu).T @ (X0 - mu) + (X1 - mu).T @ (X1 - mu)) / (2 * len(X0))
    np.testing.assert_allclose(m.covariance_, cov, rtol=0, atol=2e-6 * np.abs(cov).max())
    np.testing.assert_allclose(m._outer_gram_sum, G, rtol=0, atol=ATOL_SCALE * scale)


@pytest.mark.parametrize("stored", ["float32", "bfloat16"])
@pytest.mark.parametrize("mode,F,lag", [("bf16", 300, 7), ("bf16x2", 512, 40), ("bf16", 2048, 3)])
def test_image_path_folds_its_column_sums_too(gpu, monkeypatch, mode, F, lag, stored):
    """bf16 modes: the packed-image pre-pass reads every left frame anyway and sums it in fp64 while it packs -- no
    column-sum pass of its own (any width: the image is padded, not the sums)."""
    torch = pytest.importorskip("torch")
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", mode)
    monkeypatch.setenv("MSM_TICA_FOLD", "2")
    monkeypatch.setenv("MSM_TICA_IMG_FUSED", "0")   # the fused kernel (default up to 512 features, bf16-stored rows) has no pre-pass to fold into
    lens = [5000, 4096 + 2 * lag, lag, 2 * lag, 2 * lag + 33, 700]
    seqs = _data(F + lag, lens, F)
    if stored == "bfloat16":
        dev = [torch.from_numpy(s).cuda().to(torch.bfloat16) for s in seqs]
        seqs = [d.float().cpu().numpy() for d in dev]          # what the stored values are
    else:
        dev = [torch.from_numpy(s).cuda() for s in seqs]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = tICA(n_components=4, lag_time=lag).fit(dev)
        assert _folded(m) == 1
        monkeypatch.setenv("MSM_TICA_FOLD", "0")
        m0 = tICA(n_components=4, lag_time=lag).fit(dev)
        assert _folded(m0) == 0
    m._pull(); m0._pull()
    Cm, G, s0, st, n = _numpy_moments(seqs, lag, F)
    np.testing.assert_allclose(m._sum_0_to_TminusTau, s0, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(m._sum_tau_to_T, st, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(m.means_, m0.means_, rtol=1e-12, atol=1e-13)
    tol = (1e-3 if mode == "bf16" else 1e-5) * np.abs(G).max()   # the modes' stated accuracy on ~10,000 frames
    np.testing.assert_allclose(m._outer_gram_sum, G, rtol=0, atol=tol)
    np.testing.assert_allclose(m0._outer_gram_sum, G, rtol=0, atol=tol)      # (different shift rows: two roundings of the same sums)
    # a rejected launch is undone here as well
    bad = seqs[0].copy()
    bad[1234, 5] = np.nan
    before = m._outer_gram_sum.copy()
    monkeypatch.setenv("MSM_TICA_FOLD", "2")
    with pytest.raises(ValueError, match="NaN"):
        m.partial_fit(torch.from_numpy(bad).cuda() if stored == "float32" else torch.from_numpy(bad).cuda().to(torch.bfloat16))
    m._is_dirty = True
    m._pull()
    np.testing.assert_array_equal(m._outer_gram_sum, before)
