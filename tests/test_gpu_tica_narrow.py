"""Round 6: tica_symw_f32_kernel -- the sum/difference accumulation for F <= 256 with one workgroup
owning all of H and D (csrc/tica_symw_dev.h) -- against the numpy restatement of tica.py:401-424 (oracle/tica_oracle.py) and
against the 128-wide kernels of rounds 1-5 (MSM_TICA_SYMW=0), at every width class of the kernel family (16 / 32 / 64 / 128 /
192 / 256 columns, frame-split and block-split variants), widths that are not multiples of 4 (the piece that straddles the
row end), 1-3 features (element-wise loads), ragged / short / skipped trajectories, several K-steps and chunks, un-centred
features (mean shift), unaligned row pointers, partial_fit, and BASELINE configs[0] / configs[2] shapes."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL = 1e-6   # accumulators relative to max|G| (stated fp32 tolerance, tests/test_gpu_tica.py)


def _data(seed, lens, F, offset=2.0, scale=1.0):
    rs = np.random.RandomState(seed)
    k = min(4, F)
    M = rs.randn(k, F)
    b = rs.uniform(-offset, offset, size=F)
    out = []
    for n in lens:
        z = np.cumsum(rs.randn(n, k), axis=0) * 0.1
        out.append(((z @ M + rs.randn(n, F)) * scale + b).astype(np.float32))
    return out


def _fit(seqs, lag, k):
    from msmbuilder_amd import tICA
    from oracle.tica_oracle import TicaOracle
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = tICA(n_components=k, lag_time=lag).fit(seqs)
        o = TicaOracle(n_components=k, lag_time=lag).fit(seqs)
    return m, o


def _check(m, o, eig_rtol=1e-5):
    assert (m.n_observations_, m.n_sequences_) == (o.n_observations_, o.n_sequences_)
    assert m._lagged_symmetrised
    m._pull()
    G = o.S0 + o.Stau
    tol = ATOL * np.abs(G).max()
    np.testing.assert_allclose(m._outer_gram_sum, G, rtol=0, atol=tol)
    np.testing.assert_allclose(m._outer_0_to_T_lagged, 0.5 * (o.C + o.C.T), rtol=0, atol=tol)
    np.testing.assert_allclose(m._sum_0_to_TminusTau, o.s0, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(m._sum_tau_to_T, o.stau, rtol=1e-12, atol=1e-9)
    assert np.array_equal(m._outer_gram_sum, m._outer_gram_sum.T)
    assert np.array_equal(m._outer_0_to_T_lagged, m._outer_0_to_T_lagged.T)
    np.testing.assert_allclose(m.eigenvalues_, o.eigenvalues_, rtol=eig_rtol)


@pytest.mark.parametrize("F", [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 47, 64, 65, 100, 127, 128, 129, 171, 192, 193, 200, 255, 256])
def test_every_width_class_vs_oracle(gpu, monkeypatch, F):
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f32")
    lag = 7
    lens = [900, 301, lag, lag + 1, 2500, 5]          # ragged, == lag (skipped), one pair, several K-steps, shorter than lag
    seqs = _data(F, lens, F)
    m, o = _fit(seqs, lag, min(3, F))
    _check(m, o)
    assert m._handle is not None


@pytest.mark.parametrize("F,lag,n_seq,T", [(4, 1, 10, 9999), (171, 100, 28, 10000), (128, 100, 40, 25000), (10, 50, 30, 20000)])
def test_baseline_shapes(gpu, monkeypatch, F, lag, n_seq, T):
    """BASELINE configs[0] (10 x 9,999 x 4: dihedral sin/cos), configs[2] (28 x 10,000 x 171 contact features), configs[1]'s
    width at 1M frames, and a 10-column projection-like input: many chunks per workgroup, several slab merges."""
    import torch
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f32")
    seqs = _data(F + lag, [T] * n_seq, F, offset=1.0)
    from msmbuilder_amd import tICA
    from oracle.tica_oracle import TicaOracle
    X = torch.from_numpy(np.concatenate(seqs)).cuda()
    dev = list(X.view(n_seq, T, F).unbind(0))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = tICA(n_components=min(4, F), lag_time=lag).fit(dev)
        o = TicaOracle(n_components=min(4, F), lag_time=lag).fit(seqs)
    _check(m, o)


@pytest.mark.parametrize("F", [6, 24, 60, 120, 171, 250])
def test_equals_the_128_wide_kernels(gpu, monkeypatch, F):
    """A/B against rounds 1-5's kernels (MSM_TICA_SYMW=0: C/G form below 129 features or off multiples of 4, the 128-tile
    sum/difference kernel otherwise): same moments to the fp32 tolerance, same eigenvalues, same projection."""
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f32")
    seqs = _data(F, [3000, 1200, 4097], F)
    out = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("MSM_TICA_SYMW", sw)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = tICA(n_components=min(3, F), lag_time=11).fit(seqs)
        m._pull()
        C = m._outer_0_to_T_lagged
        out[sw] = (0.5 * (C + C.T), m._outer_gram_sum.copy(), m.eigenvalues_.copy(), m.transform(seqs[:1])[0], m._lagged_symmetrised)
    a, b = out["1"], out["0"]
    assert a[4]
    scale = np.abs(b[1]).max()
    np.testing.assert_allclose(a[0], b[0], rtol=0, atol=2 * ATOL * scale)
    np.testing.assert_allclose(a[1], b[1], rtol=0, atol=2 * ATOL * scale)
    np.testing.assert_allclose(a[2], b[2], rtol=1e-5)
    sign = np.sign((a[3] * b[3]).sum(0))
    np.testing.assert_allclose(a[3] * sign, b[3], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("F", [5, 40, 171])
def test_uncentred_features(gpu, monkeypatch, F):
    """|mean| / sigma = 100: raw fp32 moments would not even be positive definite; the kernel accumulates x - r."""
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f32")
    seqs = _data(F + 1, [4000, 4000, 1500], F, offset=100.0)
    m, o = _fit(seqs, 9, min(3, F))
    _check(m, o, eig_rtol=1e-4)
    np.testing.assert_allclose(m.covariance_, o.covariance_, rtol=0, atol=1e-5 * np.abs(o.covariance_).max())


@pytest.mark.parametrize("F", [3, 10, 171])
def test_unaligned_rows_strided_views_and_partial_fit(gpu, monkeypatch, F):
    """Row pointers at every 4-byte alignment class (views into a larger tensor starting at odd element offsets), device
    tensors, and the same data through partial_fit in pieces: identical counts, moments to tolerance."""
    import torch
    from msmbuilder_amd import tICA
    from oracle.tica_oracle import TicaOracle
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f32")
    lens = [700, 1300, 260, 999]
    seqs = _data(F + 2, lens, F)
    flat = torch.empty(sum(lens) * F + 16, dtype=torch.float32, device="cuda")
    dev, off = [], 0
    for i, s in enumerate(seqs):
        start = off + (i % 4)                       # 0, 1, 2, 3 elements past a 16-byte boundary (when F % 4 == 0 too)
        v = flat[start:start + s.size].view(s.shape)
        v.copy_(torch.from_numpy(s))
        dev.append(v)
        off = start + s.size
        off += (-off) % 4
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = tICA(n_components=min(3, F), lag_time=5).fit(dev)
        o = TicaOracle(n_components=min(3, F), lag_time=5).fit(seqs)
        p = tICA(n_components=min(3, F), lag_time=5)
        for s in seqs:
            p.partial_fit(s)
    _check(m, o)
    _check(p, o)


def test_nan_is_rejected_and_state_kept(gpu, monkeypatch):
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f32")
    seqs = _data(3, [800, 800], 20)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = tICA(n_components=2, lag_time=3).fit(seqs)
    before = m.eigenvalues_.copy()
    bad = seqs[0].copy()
    bad[17, 3] = np.nan
    with pytest.raises(ValueError):
        m.partial_fit(bad)
    np.testing.assert_array_equal(m.eigenvalues_, before)


@pytest.mark.parametrize("F", [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 47, 64, 65, 96, 97, 100, 127, 128])
def test_float64_rows_every_width_class_vs_oracle(gpu, monkeypatch, F):
    """float64 rows of up to 128 features take the same whole-matrix variants on doubles (tica_symw_f64_kernel: the fp64 matrix
    pipe; until round 6 every float64 row went to the 128-wide fp64 tile kernel, 13-17 ms per 8M frames whatever the width).
    fp64 products and sums: the oracle's moments to 1e-11, its eigenvalues to 1e-9; un-centred features (|mean| / sigma up to
    ~50: the shift row); ragged / skipped / short trajectories; odd row pointers; A/B against MSM_TICA_SYMW64=0."""
    import torch
    from msmbuilder_amd import tICA
    from oracle.tica_oracle import TicaOracle
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f32")
    lag = 7
    lens = [900, 301, lag, lag + 1, 2500, 5]
    seqs = [s.astype(np.float64) * (1.0 + 1e-9 * np.arange(s.shape[1])) for s in _data(F + 3, lens, F, offset=50.0)]
    k = min(3, F)
    o = TicaOracle(n_components=k, lag_time=lag).fit(seqs)
    G = o.S0 + o.Stau
    # device views at odd 8-byte offsets
    flat = torch.empty(sum(lens) * F + 8 * len(lens), dtype=torch.float64, device="cuda")
    dev, off = [], 0
    for i, s in enumerate(seqs):
        start = off + (i % 2)
        v = flat[start:start + s.size].view(s.shape)
        v.copy_(torch.from_numpy(s))
        dev.append(v)
        off = start + s.size
    out = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("MSM_TICA_SYMW64", sw)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = tICA(n_components=k, lag_time=lag).fit(seqs if sw == "0" else dev)
        assert (m.n_observations_, m.n_sequences_) == (o.n_observations_, o.n_sequences_)
        m._pull()
        C = np.array(m._outer_0_to_T_lagged)
        out[sw] = (C, np.array(m._outer_gram_sum), np.array(m.eigenvalues_))
        tol = 1e-11 * np.abs(G).max()
        np.testing.assert_allclose(m._outer_gram_sum, G, rtol=0, atol=tol)
        np.testing.assert_allclose(0.5 * (C + C.T), 0.5 * (o.C + o.C.T), rtol=0, atol=tol)
        np.testing.assert_allclose(m._sum_0_to_TminusTau, o.s0, rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(m._sum_tau_to_T, o.stau, rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(m.eigenvalues_, o.eigenvalues_, rtol=1e-9, atol=1e-12)
    # the whole-matrix kernel leaves the symmetrised lagged moment, the tile kernel the raw one (F = 1: both are 1 x 1)
    assert np.array_equal(out["1"][0], out["1"][0].T)
    if F > 1:
        assert not np.array_equal(out["0"][0], out["0"][0].T)


def test_float64_rows_partial_fit_mixed_with_float32(gpu, monkeypatch):
    """One handle, float32 and float64 launches in turn (both whole-matrix variants merge into the same slabs), then a wide
    float64 launch is refused as always (different width) -- counts and moments equal one float64 oracle fit."""
    from msmbuilder_amd import tICA
    from oracle.tica_oracle import TicaOracle
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f32")
    F, lag = 40, 5
    seqs32 = _data(77, [1500, 700, 2600, 900], F, offset=3.0)
    seqs = [s if i % 2 == 0 else s.astype(np.float64) for i, s in enumerate(seqs32)]
    m = tICA(n_components=3, lag_time=lag)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for s in seqs:
            m.partial_fit(s)
        o = TicaOracle(n_components=3, lag_time=lag).fit(seqs32)
    _check(m, o)
