"""GPU parity of msmbuilder_amd.preprocessing against scikit-learn's scalers (the arithmetic
msmbuilder.preprocessing wraps, preprocessing/__init__.py:56-83) and of the algebraic
StandardScaler -> tICA fold against the explicit two-step pipeline."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _seqs(seed, dtype, F=37, offset=50.0):
    rs = np.random.RandomState(seed)
    scale = rs.uniform(0.01, 30.0, size=F)
    shift = rs.uniform(-offset, offset, size=F)
    out = [(rs.randn(n, F) * scale + shift).astype(dtype) for n in (1, 4097, 300, 8, 2500)]
    out[1][:, 5] = 3.25          # constant column -> scale 1
    out[2][:, 5] = 3.25
    for a in out:
        a[:, 5] = 3.25
    return out


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_standard_scaler_matches_sklearn(gpu, dtype):
    from sklearn.preprocessing import StandardScaler as Ref
    from msmbuilder_amd.preprocessing import StandardScaler
    seqs = _seqs(0, dtype)
    ref = Ref().fit(np.concatenate(seqs))
    m = StandardScaler().fit(seqs)
    assert m.n_samples_seen_ == ref.n_samples_seen_
    np.testing.assert_allclose(m.mean_, ref.mean_, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(m.var_, ref.var_, rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(m.scale_, ref.scale_, rtol=1e-10)
    assert m.scale_[5] == 1.0
    # transform with IDENTICAL parameters is bit-exact (numpy's rounding after each in-place step)
    m.mean_, m.scale_ = ref.mean_.copy(), ref.scale_.copy()
    for X, Y in zip(seqs, m.transform(seqs)):
        assert Y.dtype == dtype and np.array_equal(Y, ref.transform(X))
    # online: partial_fit per sequence == one fit (preprocessing/base.py:182-199)
    mo = StandardScaler()
    for X in seqs:
        mo.partial_fit(X)
    np.testing.assert_allclose(mo.mean_, ref.mean_, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(mo.var_, ref.var_, rtol=1e-10, atol=1e-14)
    back = m.inverse_transform(m.transform(seqs))
    np.testing.assert_allclose(back[1], seqs[1], rtol=1e-5 if dtype == np.float32 else 1e-12, atol=1e-4 if dtype == np.float32 else 1e-10)


def test_standard_scaler_device_nan_and_inf(gpu):
    import torch
    from sklearn.preprocessing import StandardScaler as Ref
    from msmbuilder_amd.preprocessing import StandardScaler
    rs = np.random.RandomState(3)
    X = rs.randn(5000, 16).astype(np.float32) * 4 + 7
    X[rs.randint(0, 5000, 40), rs.randint(0, 16, 40)] = np.nan      # missing values are skipped
    ref = Ref().fit(X)
    Xd = torch.from_numpy(X).cuda()
    m = StandardScaler().fit([Xd[:1234], Xd[1234:]])
    assert np.array_equal(m.n_samples_seen_, ref.n_samples_seen_)
    np.testing.assert_allclose(m.mean_, ref.mean_, rtol=1e-12)
    np.testing.assert_allclose(m.var_, ref.var_, rtol=1e-10)
    Y = m.transform([Xd])[0]
    assert Y.is_cuda and Y.dtype == torch.float32
    np.testing.assert_allclose(Y.cpu().numpy(), ref.transform(X), rtol=2e-6, atol=2e-6, equal_nan=True)
    X[17, 3] = np.inf
    with pytest.raises(ValueError, match="infinity"):
        StandardScaler().fit([X])
    with pytest.raises(ValueError):
        m.transform([X[:, :5]])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_minmax_and_maxabs_match_sklearn(gpu, dtype):
    from sklearn.preprocessing import MinMaxScaler as RefMM, MaxAbsScaler as RefMA
    from msmbuilder_amd.preprocessing import MinMaxScaler, MaxAbsScaler
    seqs = _seqs(1, dtype, F=18)
    cat = np.concatenate(seqs)
    ref, m = RefMM(feature_range=(-1, 2)).fit(cat), MinMaxScaler(feature_range=(-1, 2)).fit(seqs)
    for name in ("data_min_", "data_max_", "data_range_", "scale_", "min_"):
        assert np.array_equal(getattr(m, name), getattr(ref, name)), name
    assert m.n_samples_seen_ == ref.n_samples_seen_
    for X, Y in zip(seqs, m.transform(seqs)):
        assert np.array_equal(Y, ref.transform(X))
    ref, m = RefMA().fit(cat), MaxAbsScaler().fit(seqs)
    assert np.array_equal(m.max_abs_, ref.max_abs_) and np.array_equal(m.scale_, ref.scale_)
    for X, Y in zip(seqs, m.transform(seqs)):
        assert np.array_equal(Y, ref.transform(X))


def test_fold_into_tica_equals_the_two_step_pipeline(gpu, monkeypatch):
    """StandardScaler -> tICA without writing the scaled data: same model as fitting tICA on
    scaler.transform(sequences) (float64 accumulation so that only the rounding of the scaled
    float32 copy separates the two)."""
    from msmbuilder_amd import tICA
    from msmbuilder_amd.preprocessing import StandardScaler, fold_into_tica
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f64")
    rs = np.random.RandomState(9)
    k, F = 4, 20
    M = rs.randn(k, F)
    a = np.exp(-1.0 / np.array([40.0, 15.0, 6.0, 3.0]))
    seqs = []
    for n in (3000, 4500, 1200):
        z = np.zeros((n, k))
        e = rs.randn(n, k)
        for t in range(1, n):
            z[t] = a * z[t - 1] + np.sqrt(1 - a * a) * e[t]
        seqs.append(((z.dot(M) + 0.3 * rs.randn(n, F)) * rs.uniform(0.1, 20, F) + rs.uniform(-5, 5, F)).astype(np.float64))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc = StandardScaler().fit(seqs)
        two = tICA(n_components=3, lag_time=5).fit(sc.transform(seqs))
        one = fold_into_tica(StandardScaler(), tICA(n_components=3, lag_time=5), seqs)
    np.testing.assert_allclose(one.means_, two.means_, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(one.covariance_, two.covariance_, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(one.offset_correlation_, two.offset_correlation_, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(one.eigenvalues_, two.eigenvalues_, rtol=1e-9)
    assert one.shrinkage_ == pytest.approx(two.shrinkage_, rel=1e-8)
    Y1, Y2 = one.transform(seqs), two.transform(sc.transform(seqs))     # raw rows in, same projection out
    for y1, y2 in zip(Y1, Y2):
        s = np.sign(np.sum(y1 * y2, axis=0))
        np.testing.assert_allclose(y1 * s, y2, rtol=1e-7, atol=1e-8)
    assert one.score(seqs) == pytest.approx(two.score(sc.transform(seqs)), rel=1e-9)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_column_order_statistics_exact(gpu, dtype):
    from msmbuilder_amd.preprocessing import column_order_statistics
    rs = np.random.RandomState(4)
    X = (rs.randn(7001, 13) * rs.uniform(1e-3, 1e3, 13) - 2).astype(dtype)
    X[:, 3] = np.round(X[:, 3])                      # heavy ties
    X[:, 7] = -0.0
    X[5, 7] = 0.0
    X[rs.randint(0, 7001, 30), 2] = np.nan           # missing values are not ranked
    srt = np.sort(X, axis=0)                         # NaN sorts last
    n = (~np.isnan(X)).sum(0)
    ranks = np.stack([np.zeros(13, np.int64), n // 2, n - 1, rs.randint(0, n.min(), 13), np.full(13, -1)])
    got = column_order_statistics([X[:1000], X[1000:1001], X[1001:]], ranks)
    assert got.dtype == dtype
    for r in range(4):
        assert np.array_equal(got[r], srt[ranks[r], np.arange(13)]), r
    assert np.isnan(got[4]).all()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_robust_scaler_matches_sklearn(gpu, dtype):
    import torch
    from sklearn.preprocessing import RobustScaler as Ref
    from msmbuilder_amd.preprocessing import RobustScaler
    for seed, kw in ((0, {}), (1, dict(quantile_range=(10.0, 90.0), unit_variance=True)), (2, dict(with_centering=False))):
        seqs = _seqs(seed, dtype, F=21)
        seqs[1][7, 2] = np.nan
        cat = np.concatenate(seqs)
        ref, m = Ref(**kw).fit(cat), RobustScaler(**kw).fit(seqs)
        if ref.center_ is None:
            assert m.center_ is None
        else:
            assert m.center_.dtype == ref.center_.dtype and np.array_equal(m.center_, ref.center_)
        assert m.scale_.dtype == ref.scale_.dtype and np.array_equal(m.scale_, ref.scale_)
        for X, Y in zip(seqs, m.transform(seqs)):
            assert Y.dtype == dtype and np.array_equal(Y, ref.transform(X), equal_nan=True)
        md = RobustScaler(**kw).fit([torch.from_numpy(s).cuda() for s in seqs])
        assert np.array_equal(md.scale_, ref.scale_)
    even = [np.arange(10, dtype=dtype).reshape(10, 1)[::-1].copy()]      # even count: mean of the middle pair
    assert np.array_equal(RobustScaler().fit(even).center_, Ref().fit(even[0]).center_)


def test_colstats_and_scale_apply_strided_host_rows(gpu):
    """C ABI with a row pitch larger than n_features on HOST memory (the Python wrappers always pass contiguous arrays):
    the staging copy has to be a 2-D one."""
    import ctypes as C
    L = gpu.lib()
    rs = np.random.RandomState(4)
    F, ld = 37, 48
    blocks = [rs.randn(n, ld).astype(np.float32) * 3 + 1 for n in (700, 1, 2600)]
    ptrs = (C.c_void_p * 3)(*[b.ctypes.data for b in blocks])
    rows = (C.c_int64 * 3)(*[len(b) for b in blocks])
    out = np.empty((5, F))
    has_inf = C.c_int(0)
    gpu.check(L.msm_colstats(ptrs, rows, 3, 4, F, ld, 0, out.ctypes.data, C.byref(has_inf)))
    X = np.concatenate([b[:, :F] for b in blocks]).astype(np.float64)
    assert has_inf.value == 0 and np.all(out[0] == len(X))
    np.testing.assert_allclose(out[1], X.mean(0), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out[2], ((X - X.mean(0)) ** 2).sum(0), rtol=1e-10)
    np.testing.assert_array_equal(out[3], X.min(0))
    np.testing.assert_array_equal(out[4], X.max(0))
    shift, scale = X.mean(0), X.std(0)
    b = blocks[2]
    res = np.full((len(b), 40), -7.0, dtype=np.float32)
    gpu.check(L.msm_scale_apply(C.c_void_p(b.ctypes.data), 4, len(b), F, ld, C.c_void_p(shift.ctypes.data),
                                C.c_void_p(scale.ctypes.data), 0, C.c_void_p(res.ctypes.data), 40, 0))
    want = (b[:, :F].astype(np.float64) - shift).astype(np.float32)     # numpy's in-place X -= mean; X /= scale on float32:
    want = (want.astype(np.float64) / scale).astype(np.float32)         # each step rounded to the array dtype
    np.testing.assert_array_equal(res[:, :F], want)
    assert np.all(res[:, F:] == -7.0)
