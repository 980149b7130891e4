"""Round 6: ALL 14 unprefixed reference-signature exports of libmsmhip.so (include/msmhip_libdistance.h =
/root/reference/msmbuilder/libdistance/libdistance.pyx:26-67, name for name) called straight through ctypes -- no Python
wrapper of this package in between -- with the SAME argument lists as the reference's own functions compiled from its
headers (oracle/_ref/libref_libdistance.so exports them as ref_<name>; where that library is absent the plain-C
restatement oracle/liboracle_libdistance.so stands in).  Both element types, X_indices NULL and non-NULL, every metric;
array outputs bit for bit, returned sums (inertia, sumdist) to 1e-13 (an fp64 tree sum instead of a sequential one, as the
header says); an unknown metric returns -1 / leaves `out` untouched (assign.hpp:15-18, dist.hpp:11-14).
A swapped argument in csrc/libdistance_compat.hip cannot pass this file."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

METRICS = ["euclidean", "sqeuclidean", "cityblock", "chebyshev", "canberra", "braycurtis", "hamming", "jaccard"]
_i = C.c_int64
_p = C.c_void_p


@pytest.fixture(scope="module")
def checker():
    """(call(name, *args) on the checker library, is_reference)"""
    from oracle.libdistance_oracle import Ref, Oracle
    if Ref.available():
        lib = Ref().lib
        return (lambda name: getattr(lib, "ref_" + name)), True
    o = Oracle()
    return o, False


def _data(dtype, n=157, m=9, nb=11, seed=0, metric="euclidean"):
    rs = np.random.RandomState(seed)
    if metric in ("hamming", "jaccard"):          # categorical-ish values so that equal / zero entries occur
        X = rs.randint(0, 3, (n, m)).astype(dtype)
        Y = rs.randint(0, 3, (nb, m)).astype(dtype)
    else:
        X = (rs.randn(n, m) * 2).astype(dtype)
        Y = (rs.randn(nb, m) * 2).astype(dtype)
        X[3] = Y[2]                                # an exact hit
        X[5, :] = 0.0                              # canberra / braycurtis: 0 / 0 terms
    idx = np.ascontiguousarray(rs.permutation(n)[: n // 3], dtype=np.int64)
    return np.ascontiguousarray(X), np.ascontiguousarray(Y), idx


def _sfx(dtype):
    return "double" if dtype == np.float64 else "float"


def _ref_call(checker, name, restype, args):
    get, is_ref = checker
    assert is_ref
    fn = get(name)
    fn.restype = restype
    return fn(*args)


def _mine(gpu, name, restype):
    fn = getattr(gpu.lib(), name)
    fn.restype = restype
    return fn


@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("with_idx", [False, True])
def test_assign_nearest(gpu, checker, metric, dtype, with_idx):
    X, Y, idx = _data(dtype, metric=metric)
    n_out = len(idx) if with_idx else len(X)
    name = "assign_nearest_" + _sfx(dtype)

    def args(out):
        return (_p(X.ctypes.data), _p(Y.ctypes.data), metric.encode(), _p(idx.ctypes.data) if with_idx else None,
                _i(X.shape[0]), _i(Y.shape[0]), _i(X.shape[1]), _i(n_out), _p(out.ctypes.data))
    out = np.full(n_out, -7, dtype=np.int64)
    inertia = _mine(gpu, name, C.c_double)(*args(out))
    if checker[1]:
        want = np.full(n_out, -9, dtype=np.int64)
        want_inertia = _ref_call(checker, name, C.c_double, args(want))
    else:
        want, want_inertia = checker[0].assign_nearest(X, Y, metric, idx if with_idx else None)
    np.testing.assert_array_equal(out, want)
    assert abs(inertia - want_inertia) <= 1e-13 * abs(want_inertia)
    # unknown metric: -1, assignments untouched
    out2 = np.full(n_out, -7, dtype=np.int64)
    a = list(args(out2))
    a[2] = b"no-such-metric"
    assert _mine(gpu, name, C.c_double)(*a) == -1.0
    assert np.all(out2 == -7)


@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_dist_and_dist_X_indices(gpu, checker, metric, dtype):
    X, Y, idx = _data(dtype, seed=1, metric=metric)
    y = np.ascontiguousarray(Y[4])
    for with_idx in (False, True):
        name = "dist_" + _sfx(dtype) + ("_X_indices" if with_idx else "")
        n_out = len(idx) if with_idx else len(X)

        def args(out, met=metric):
            base = [_p(X.ctypes.data), _p(y.ctypes.data), met.encode(), _i(X.shape[0]), _i(X.shape[1])]
            if with_idx:
                base += [_p(idx.ctypes.data), _i(len(idx))]
            return tuple(base + [_p(out.ctypes.data)])
        out = np.full(n_out, -7.0)
        _mine(gpu, name, None)(*args(out))
        if checker[1]:
            want = np.full(n_out, -9.0)
            _ref_call(checker, name, None, args(want))
        else:
            want = checker[0].dist(X, y, metric, idx if with_idx else None)
        assert np.array_equal(out, want, equal_nan=True)
        out2 = np.full(n_out, -7.0)
        _mine(gpu, name, None)(*args(out2, "no-such-metric"))
        assert np.all(out2 == -7.0)               # dist.hpp:11-14: returns without touching `out`


@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cdist(gpu, checker, metric, dtype):
    X, Y, _ = _data(dtype, seed=2, metric=metric)
    name = "cdist_" + _sfx(dtype)

    def args(out, met=metric):
        return (_p(X.ctypes.data), _p(Y.ctypes.data), met.encode(), _i(X.shape[0]), _i(Y.shape[0]), _i(X.shape[1]), _p(out.ctypes.data))
    out = np.full((X.shape[0], Y.shape[0]), -7.0)
    _mine(gpu, name, None)(*args(out))
    if checker[1]:
        want = np.full(out.shape, -9.0)
        _ref_call(checker, name, None, args(want))
    else:
        want = checker[0].cdist(X, Y, metric)
    assert np.array_equal(out, want, equal_nan=True)
    out2 = np.full(out.shape, -7.0)
    _mine(gpu, name, None)(*args(out2, "no-such-metric"))
    assert np.all(out2 == -7.0)


@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_pdist_and_pdist_X_indices(gpu, checker, metric, dtype):
    X, _, idx = _data(dtype, n=61, seed=3, metric=metric)
    for with_idx in (False, True):
        name = "pdist_" + _sfx(dtype) + ("_X_indices" if with_idx else "")
        k = len(idx) if with_idx else len(X)
        n_out = k * (k - 1) // 2

        def args(out, met=metric):
            base = [_p(X.ctypes.data), met.encode(), _i(X.shape[0]), _i(X.shape[1])]
            if with_idx:
                base += [_p(idx.ctypes.data), _i(len(idx))]
            return tuple(base + [_p(out.ctypes.data)])
        out = np.full(n_out, -7.0)
        _mine(gpu, name, None)(*args(out))
        if checker[1]:
            want = np.full(n_out, -9.0)
            _ref_call(checker, name, None, args(want))
        else:
            want = checker[0].pdist(X, metric, idx if with_idx else None)
        assert np.array_equal(out, want, equal_nan=True)
        out2 = np.full(n_out, -7.0)
        _mine(gpu, name, None)(*args(out2, "no-such-metric"))
        assert np.all(out2 == -7.0)


@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_sumdist(gpu, checker, metric, dtype):
    X, _, _ = _data(dtype, n=83, seed=4, metric=metric)
    rs = np.random.RandomState(5)
    pairs = np.ascontiguousarray(rs.randint(0, len(X), (200, 2)), dtype=np.int64)
    name = "sumdist_" + _sfx(dtype)

    def args(met=metric):
        return (_p(X.ctypes.data), met.encode(), _i(X.shape[0]), _i(X.shape[1]), _p(pairs.ctypes.data), _i(len(pairs)))
    got = _mine(gpu, name, C.c_double)(*args())
    want = _ref_call(checker, name, C.c_double, args()) if checker[1] else checker[0].sumdist(X, metric, pairs)
    if np.isnan(want):
        assert np.isnan(got)
    else:
        assert abs(got - want) <= 1e-13 * abs(want)
    assert _mine(gpu, name, C.c_double)(*args("no-such-metric")) == -1.0


def test_every_declared_compat_symbol_is_called_here(gpu):
    """The list this file exercises is the header's list (parsed from include/msmhip_libdistance.h)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "msmhip_libdistance.h")).read()
    names = set(re.findall(r"^(?:double|void)\s+(\w+)\(", text, flags=re.M))
    covered = {b + "_" + t + s for b, sfxs in (("assign_nearest", ("",)), ("dist", ("", "_X_indices")), ("cdist", ("",)),
                                              ("pdist", ("", "_X_indices")), ("sumdist", ("",)))
               for t in ("double", "float") for s in sfxs}
    assert names == covered and len(names) == 14
    for n in names:
        assert hasattr(gpu.lib(), n)
