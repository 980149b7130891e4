"""GPU parity of msmbuilder_amd.libdistance / KCenters: bit-exact against the oracle
(oracle/libdistance_oracle.c) and the golden vectors of the compiled reference."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

METRICS = ("euclidean", "sqeuclidean", "cityblock", "chebyshev", "canberra", "braycurtis",
           "hamming", "jaccard")


@pytest.fixture(scope="module")
def oracle():
    from oracle.libdistance_oracle import Oracle
    return Oracle()


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("metric", METRICS)
def test_golden_vectors(gpu, golden_dir, metric, dtype):
    from msmbuilder_amd import libdistance as ld
    g = np.load(os.path.join(golden_dir, "libdistance_golden.npz"))
    dn = "f32" if dtype == np.float32 else "f64"
    idx = g["idx"]
    for tag, (A, B) in (("g", (g["X"], g["Y"])), ("r", (g["Xr"], g["Yr"]))):
        A, B = A.astype(dtype), B.astype(dtype)
        p = "%s_%s_%s_" % (metric, dn, tag)
        assert _same(ld.cdist(A, B, metric), g[p + "cdist"])
        lab, inertia = ld.assign_nearest(A, B, metric)
        assert lab.dtype == np.intp and isinstance(inertia, float)
        assert np.array_equal(lab, g[p + "assign"])
        gi = float(g[p + "inertia"])
        assert (inertia == gi) or (np.isnan(gi) and np.isnan(inertia)) or abs(inertia - gi) <= 1e-13 * abs(gi)
        lab, inertia = ld.assign_nearest(A, B, metric, idx)
        assert np.array_equal(lab, g[p + "assign_idx"])
        assert _same(ld.dist(A, B[2], metric), g[p + "dist"])
        assert _same(ld.dist(A, B[2], metric, idx), g[p + "dist_idx"])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("n,k,f", [(1, 1, 1), (300, 9, 3), (1000, 37, 10), (777, 200, 45), (513, 17, 130),
                                   # wide-row streaming path (16-byte aligned rows longer than one chunk)
                                   (777, 200, 44), (513, 17, 132), (2100, 8, 64), (300, 1, 36), (70000, 3, 40),
                                   # two rows per lane (more than 32 centres): several 512-row tiles, partial last chunk / group
                                   (1500, 70, 132), (1030, 33, 48),
                                   # centre groups sized to the shape: K <= 8 (8-centre groups), float64 with K > 16 (32-centre
                                   # groups: one full group, a partial one, several)
                                   (600, 5, 260), (520, 32, 48), (520, 31, 48), (900, 100, 256)])
def test_bit_exact_vs_oracle(gpu, oracle, metric, dtype, n, k, f):
    from msmbuilder_amd import libdistance as ld
    rs = np.random.RandomState(n + k + f)
    X = rs.randn(n, f).astype(dtype)
    Y = rs.randn(k, f).astype(dtype)
    if metric in ("hamming", "jaccard"):
        X, Y = np.round(X).astype(dtype), np.round(Y).astype(dtype)
    Y[: min(k, 3)] = X[: min(k, 3)]          # exact hits / duplicate-distance ties
    if k > 4:
        Y[4] = Y[1]                           # identical centres: lowest index must win
    idx = rs.randint(0, n, size=41).astype(np.int64)
    with np.errstate(all="ignore"):
        lab, inertia = ld.assign_nearest(X, Y, metric)
        lab_o, inertia_o, mind_o = oracle.assign_nearest(X, Y, metric, return_distances=True)
        assert np.array_equal(lab, lab_o)
        assert (np.isnan(inertia) and np.isnan(inertia_o)) or abs(inertia - inertia_o) <= 1e-13 * abs(inertia_o)
        lab, _ = ld.assign_nearest(X, Y, metric, idx)
        assert np.array_equal(lab, oracle.assign_nearest(X, Y, metric, idx)[0])
        assert _same(ld.cdist(X, Y, metric), oracle.cdist(X, Y, metric))
        assert _same(ld.dist(X, Y[0], metric), oracle.dist(X, Y[0], metric))
        assert _same(ld.dist(X, Y[0], metric, idx), oracle.dist(X, Y[0], metric, idx))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("metric", METRICS)
def test_pdist_sumdist_bit_exact(gpu, oracle, metric, dtype):
    """'next' rows of SURVEY 8(f3): pdist / sumdist (pdist.hpp, sumdist.hpp; reference tests
    tests/test_libdistance.py:128-148, 212-219)."""
    import scipy.spatial.distance
    from msmbuilder_amd import libdistance as ld
    rs = np.random.RandomState(11)
    for n, f in ((2, 1), (37, 5), (300, 33), (64, 1100)):
        X = rs.randn(n, f).astype(dtype)
        if metric in ("hamming", "jaccard"):
            X = np.round(X).astype(dtype)
        idx = rs.randint(0, n, size=min(n, 11)).astype(np.int64)
        pairs = rs.randint(0, n, size=(29, 2)).astype(np.int64)
        with np.errstate(all="ignore"):
            got = ld.pdist(X, metric)
            assert got.shape == (n * (n - 1) // 2,) and got.dtype == np.float64
            assert _same(got, oracle.pdist(X, metric))
            assert _same(ld.pdist(X, metric, idx), oracle.pdist(X, metric, idx))
            a, b = ld.sumdist(X, metric, pairs), oracle.sumdist(X, metric, pairs)
            assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-13 * abs(b)
            if metric != "jaccard" and f > 1:
                np.testing.assert_almost_equal(got, scipy.spatial.distance.pdist(X, metric),
                                               decimal=4 if dtype == np.float32 else 9)
    with pytest.raises(ValueError):
        ld.sumdist(X, metric, np.zeros((3, 3), dtype=np.int64))
    assert ld.pdist(np.zeros((1, 3), dtype), metric).shape == (0,)


def test_reference_test_contract(gpu):
    """Restated from the reference's tests/test_libdistance.py:28-73,115-196."""
    import scipy.spatial.distance
    from msmbuilder_amd.libdistance import assign_nearest, cdist, dist
    rs = np.random.RandomState(0)
    Xd, Yd = rs.randn(10, 2), rs.randn(3, 2)
    Xf, Yf = rs.randn(10, 2).astype(np.float32), rs.randn(3, 2).astype(np.float32)
    idx = rs.randint(0, 10, size=5).astype(np.intp)
    for metric in METRICS:
        for X, Y in ((Xd, Yd), (Xf, Yf)):
            dec = 5 if X.dtype == np.float32 else 10
            c = cdist(X, Y, metric)
            assert c.shape == (10, 3)
            if metric != "jaccard":  # scipy >= 1.15 booleanises jaccard inputs; the reference does not
                np.testing.assert_almost_equal(c, scipy.spatial.distance.cdist(X, Y, metric), decimal=dec)
            if not (metric == "canberra" and X.dtype == np.float32):
                a, inertia = assign_nearest(X, Y, metric)
                np.testing.assert_array_equal(a, c.argmin(axis=1))
                np.testing.assert_almost_equal(inertia, c[np.arange(10), a].sum(), decimal=dec)
                a, inertia = assign_nearest(X, Y, metric, idx)
                c2 = cdist(X[idx], Y, metric)
                np.testing.assert_array_equal(a, c2.argmin(axis=1))
                np.testing.assert_almost_equal(inertia, c2[np.arange(5), a].sum(), decimal=dec)
            np.testing.assert_almost_equal(dist(X, Y[0], metric), c[:, 0], decimal=dec)
            np.testing.assert_almost_equal(dist(X, Y[0], metric, idx), c[idx, 0], decimal=dec)


def test_error_contract(gpu):
    from msmbuilder_amd.libdistance import assign_nearest, cdist, dist
    X, Y = np.zeros((4, 2), np.float32), np.zeros((2, 2), np.float32)
    with pytest.raises(ValueError):
        assign_nearest(X, Y, "minkowski")
    with pytest.raises(ValueError):
        assign_nearest(X, Y, "rmsd")
    with pytest.raises(TypeError):
        assign_nearest(X, Y.astype(np.float64), "euclidean")
    with pytest.raises(TypeError):
        cdist(X.astype(np.int32), Y.astype(np.int32), "euclidean")
    with pytest.raises(ValueError):
        cdist(X, np.zeros((2, 3), np.float32), "euclidean")
    with pytest.raises(ValueError):
        dist(X[:, ::-1], Y[0], "euclidean")     # not C-contiguous
    lab, inertia = assign_nearest(np.zeros((0, 2), np.float32), Y, "euclidean")
    assert lab.shape == (0,) and inertia == 0.0


def test_compat_symbols_match(gpu, oracle):
    """The unprefixed reference-signature entry points (include/msmhip_libdistance.h)."""
    import ctypes as C
    L = gpu.lib()
    rs = np.random.RandomState(1)
    X, Y = rs.randn(100, 6).astype(np.float32), rs.randn(7, 6).astype(np.float32)
    out = np.zeros(100, dtype=np.int64)
    L.assign_nearest_float.restype = C.c_double
    inertia = L.assign_nearest_float(C.c_void_p(X.ctypes.data), C.c_void_p(Y.ctypes.data), b"cityblock", None,
                                     C.c_int64(100), C.c_int64(7), C.c_int64(6), C.c_int64(100),
                                     C.c_void_p(out.ctypes.data))
    lab_o, inertia_o = oracle.assign_nearest(X, Y, "cityblock")
    assert np.array_equal(out, lab_o) and abs(inertia - inertia_o) < 1e-12 * inertia_o
    assert L.assign_nearest_float(C.c_void_p(X.ctypes.data), C.c_void_p(Y.ctypes.data), b"nope", None,
                                  C.c_int64(100), C.c_int64(7), C.c_int64(6), C.c_int64(100),
                                  C.c_void_p(out.ctypes.data)) == -1.0
    d = np.zeros((100, 7))
    L.cdist_float(C.c_void_p(X.ctypes.data), C.c_void_p(Y.ctypes.data), b"euclidean", C.c_int64(100),
                  C.c_int64(7), C.c_int64(6), C.c_void_p(d.ctypes.data))
    assert np.array_equal(d, oracle.cdist(X, Y, "euclidean"))


# ------------------------------------------------------------------ KCenters
def test_kcenters_golden(gpu, golden_dir):
    from msmbuilder_amd import KCenters
    g = np.load(os.path.join(golden_dir, "kcenters_golden.npz"))
    rs = np.random.RandomState(1)
    seqs = [rs.randn(23, 2).astype(np.float32), rs.randn(10, 2).astype(np.float32)]
    m = KCenters(n_clusters=3, random_state=0).fit(seqs)
    assert m.cluster_ids_ == [0, 21, 16] == list(g["K1_ids"])           # SURVEY.md known answer
    assert m.inertia_ == 29.00724663036992 == float(g["K1_inertia"])
    assert np.array_equal(np.concatenate(m.labels_), g["K1_labels"])
    assert np.array_equal(np.concatenate(m.distances_), g["K1_distances"])
    assert np.array_equal(m.cluster_centers_, g["K1_centers"]) and m.cluster_centers_.dtype == np.float32
    assert np.array_equal(np.concatenate(m.predict(seqs)), g["K1_predict"])
    assert m.labels_[1].dtype == np.int64 and m.distances_[0].dtype == np.float64
    Xk = [g["K2_seq%d" % i] for i in range(3)]
    for metric in ("euclidean", "sqeuclidean", "cityblock", "chebyshev", "canberra", "braycurtis"):
        for dt, dn in ((np.float32, "f32"), (np.float64, "f64")):
            s = [x.astype(dt) for x in Xk]
            m = KCenters(n_clusters=12, metric=metric, random_state=3).fit(s)
            p = "K2_%s_%s_" % (metric, dn)
            assert m.cluster_ids_ == list(g[p + "ids"]), (metric, dn)
            assert np.array_equal(np.concatenate(m.labels_), g[p + "labels"])
            assert np.array_equal(np.concatenate(m.distances_), g[p + "distances"])
            assert m.inertia_ == float(g[p + "inertia"])
            assert np.array_equal(np.concatenate(m.predict(s)), g[p + "predict"])
            assert m.summarize() == str(g[p + "summarize"])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,f,k", [(5000, 10, 50), (70000, 10, 200), (3000, 171, 20), (1025, 40, 1025), (3000, 172, 20),
                                   (140000, 36, 30)])
def test_kcenters_vs_oracle(gpu, oracle, n, f, k, dtype):
    from msmbuilder_amd import KCenters
    rs = np.random.RandomState(n)
    X = rs.randn(n, f).astype(dtype)
    X[100:110] = X[5]       # duplicates -> zero distances and argmax ties
    m = KCenters(n_clusters=k, random_state=7).fit([X[: n // 2], X[n // 2:]])
    ids, labels, dist = oracle.kcenters_fit(X, k, "euclidean", m.cluster_ids_[0])
    assert m.cluster_ids_ == list(ids)
    assert np.array_equal(np.concatenate(m.labels_), labels)
    assert np.array_equal(np.concatenate(m.distances_), dist)
    assert m.inertia_ == np.sum(dist)


def test_kcenters_reference_tests(gpu):
    """Restated from the reference's tests/test_kcenters.py:11-106."""
    import scipy.spatial.distance
    from msmbuilder_amd import KCenters, libdistance
    m = KCenters(n_clusters=3)
    m.fit([np.random.randn(23, 2), np.random.randn(10, 2)])
    assert isinstance(m.labels_, list) and isinstance(m.distances_, list) and len(m.labels_) == 2
    assert m.cluster_centers_.shape == (3, 2)
    assert m.labels_[0].shape == (23,) and m.labels_[1].shape == (10,)
    assert m.distances_[0].shape == (23,) and m.distances_[1].shape == (10,)
    assert m.fit_predict([np.random.randn(10, 2)])[0].shape == (10,)
    data = [np.zeros((10, 2)), np.ones((10, 2)), 0.5 * np.ones((10, 2))]
    m = KCenters(n_clusters=2, random_state=0).fit(data)
    assert np.all(m.cluster_centers_ == np.array([[0, 0], [1, 1]])) or \
        np.all(m.cluster_centers_ == np.array([[1, 1], [0, 0]]))
    np.testing.assert_allclose(np.unique(np.concatenate(m.distances_)), [0, np.sqrt(2) / 2])
    for metric in ("euclidean", "cityblock"):
        model = KCenters(n_clusters=10, metric=metric)
        data = np.random.randn(100, 2)
        l1, l2 = model.fit_predict([data]), model.predict([data])
        assert np.array_equal(l1[0], l2[0])
        pairs = scipy.spatial.distance.cdist(data, model.cluster_centers_, metric=metric)
        assert np.array_equal(l2[0], np.argmin(pairs, axis=1))
    data = np.random.RandomState(0).randn(100, 2)
    a = KCenters(n_clusters=10, random_state=0, metric='euclidean').fit_predict([data])[0]
    b = KCenters(n_clusters=10, random_state=0, metric='sqeuclidean').fit_predict([data])[0]
    assert np.array_equal(a, b)
    X = np.random.RandomState(1).randn(100, 2)
    X32, X64 = X.astype(np.float32), X.astype(np.float64)
    m1 = KCenters(n_clusters=10, random_state=0).fit([X32])
    m2 = KCenters(n_clusters=10, random_state=0).fit([X64])
    np.testing.assert_allclose(m1.cluster_centers_, m2.cluster_centers_, rtol=1e-6)
    np.testing.assert_allclose(m1.distances_[0], m2.distances_[0], rtol=1e-5, atol=1e-6)
    assert np.array_equal(m1.labels_[0], m2.labels_[0])
    assert np.array_equal(m1.predict([X32])[0], m1.labels_[0])
    np.testing.assert_almost_equal(float(m1.inertia_),
                                   libdistance.assign_nearest(X32, m1.cluster_centers_, "euclidean")[1])
    # integer input is promoted to float64 (kcenters.py:80-82), labels are intp
    mi = KCenters(n_clusters=4, random_state=0).fit([np.arange(40).reshape(20, 2)])
    assert mi.cluster_centers_.dtype == np.float64 and mi.predict([np.arange(40).reshape(20, 2)])[0].dtype == np.intp


def test_kcenters_device_resident(gpu, oracle):
    torch = pytest.importorskip("torch")
    from msmbuilder_amd import KCenters
    rs = np.random.RandomState(2)
    X = rs.randn(20000, 10)
    m = KCenters(n_clusters=40, random_state=1).fit([torch.from_numpy(X[:9000]).cuda(), torch.from_numpy(X[9000:]).cuda()])
    ids, labels, dist = oracle.kcenters_fit(X, 40, "euclidean", m.cluster_ids_[0])
    assert m.cluster_ids_ == list(ids)
    assert m.labels_[0].is_cuda and m.distances_[1].is_cuda
    assert np.array_equal(torch.cat(m.labels_).cpu().numpy(), labels)
    assert np.array_equal(torch.cat(m.distances_).cpu().numpy(), dist)
    assert abs(m.inertia_ - dist.sum()) <= 1e-12 * dist.sum()
    lab = m.predict([torch.from_numpy(X).cuda()])[0]
    assert np.array_equal(lab.cpu().numpy(), oracle.assign_nearest(X, X[ids], "euclidean")[0])
