"""oracle/libdistance_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-ends for the two CPU checkers of the libdistance path:

* ``Oracle``  -> oracle/liboracle_libdistance.so, our plain-C restatement
  (oracle/libdistance_oracle.c) of /root/reference/msmbuilder/libdistance/src.
* ``Ref``     -> oracle/_ref/libref_libdistance.so, the reference's own headers
  compiled by oracle/Makefile (present when built in the dev container; it
  travels to the GPU box as a prebuilt file).

Both expose the python-level signatures of the reference's
``msmbuilder.libdistance`` module (libdistance.pyx:82-270): ``assign_nearest``,
``cdist``, ``dist`` with the same argument order, dtype rules and error types.
Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg
import this module.  Parity status: pinned (see libdistance_oracle.c header).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle_libdistance.so")
REF_SO = os.path.join(_HERE, "_ref", "libref_libdistance.so")

VECTOR_METRICS = ("euclidean", "sqeuclidean", "cityblock", "chebyshev",
                  "canberra", "braycurtis", "hamming", "jaccard", "cityblock")

_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)


def build(force: bool = False) -> None:
    """Compile the C restatement (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(ORACLE_SO) or (
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(_HERE, "libdistance_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    elif not os.path.exists(REF_SO) and os.path.isdir("/root/reference/msmbuilder/libdistance/src"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])


def _ptr(a, typ):
    return a.ctypes.data_as(typ) if a is not None else None


def _check(X, Y, metric):
    if not (isinstance(X, np.ndarray) and isinstance(Y, np.ndarray)):
        raise TypeError()
    if metric not in VECTOR_METRICS:
        raise ValueError('metric must be one of %s' % ', '.join("'%s'" % s for s in VECTOR_METRICS))
    if X.dtype == np.float64 and Y.dtype == np.float64:
        return "f64"
    if X.dtype == np.float32 and Y.dtype == np.float32:
        return "f32"
    raise TypeError('X and y must be both float32 or float64')


class Oracle:
    """Our C restatement."""

    def __init__(self):
        build()
        self.lib = C.CDLL(ORACLE_SO)
        self.lib.oracle_assign_nearest_f64.restype = C.c_double
        self.lib.oracle_assign_nearest_f32.restype = C.c_double
        self.lib.oracle_sumdist_f64.restype = C.c_double
        self.lib.oracle_sumdist_f32.restype = C.c_double

    def assign_nearest(self, X, Y, metric, X_indices=None, return_distances=False):
        kind = _check(X, Y, metric)
        X = np.ascontiguousarray(X)
        Y = np.ascontiguousarray(Y)
        assert X.shape[1] == Y.shape[1]
        idx = None if X_indices is None else np.ascontiguousarray(X_indices, dtype=np.int64)
        n = X.shape[0] if idx is None else idx.shape[0]
        out = np.zeros(n, dtype=np.intp)
        mind = np.zeros(n, dtype=np.float64)
        fn = getattr(self.lib, "oracle_assign_nearest_" + kind)
        inertia = fn(C.c_void_p(X.ctypes.data), C.c_void_p(Y.ctypes.data), metric.encode(),
                     _ptr(idx, _i64p), C.c_int64(X.shape[0]), C.c_int64(Y.shape[0]),
                     C.c_int64(X.shape[1]), C.c_int64(n), _ptr(out, _i64p), _ptr(mind, _f64p))
        if return_distances:
            return out, inertia, mind
        return out, inertia

    def cdist(self, XA, XB, metric):
        kind = _check(XA, XB, metric)
        XA = np.ascontiguousarray(XA)
        XB = np.ascontiguousarray(XB)
        if XA.shape[1] != XB.shape[1]:
            raise ValueError('XA and XB must have the same number of columns')
        out = np.zeros((XA.shape[0], XB.shape[0]), dtype=np.float64)
        getattr(self.lib, "oracle_cdist_" + kind)(
            C.c_void_p(XA.ctypes.data), C.c_void_p(XB.ctypes.data), metric.encode(),
            C.c_int64(XA.shape[0]), C.c_int64(XB.shape[0]), C.c_int64(XA.shape[1]), _ptr(out, _f64p))
        return out

    def dist(self, X, y, metric, X_indices=None):
        kind = _check(X, y, metric)
        X = np.ascontiguousarray(X)
        y = np.ascontiguousarray(y)
        assert X.shape[1] == y.shape[0]
        idx = None if X_indices is None else np.ascontiguousarray(X_indices, dtype=np.int64)
        n = X.shape[0] if idx is None else idx.shape[0]
        out = np.zeros(n, dtype=np.float64)
        getattr(self.lib, "oracle_dist_" + kind)(
            C.c_void_p(X.ctypes.data), C.c_void_p(y.ctypes.data), metric.encode(),
            C.c_int64(X.shape[0]), C.c_int64(X.shape[1]), _ptr(idx, _i64p), C.c_int64(n),
            _ptr(out, _f64p))
        return out

    def pdist(self, X, metric, X_indices=None):
        kind = _check(X, X, metric)
        X = np.ascontiguousarray(X)
        idx = None if X_indices is None else np.ascontiguousarray(X_indices, dtype=np.int64)
        n = X.shape[0] if idx is None else idx.shape[0]
        out = np.zeros(n * (n - 1) // 2, dtype=np.float64)
        getattr(self.lib, "oracle_pdist_" + kind)(
            C.c_void_p(X.ctypes.data), metric.encode(), C.c_int64(X.shape[0]), C.c_int64(X.shape[1]),
            _ptr(idx, _i64p), C.c_int64(n), _ptr(out, _f64p))
        return out

    def sumdist(self, X, metric, pair_indices):
        kind = _check(X, X, metric)
        X = np.ascontiguousarray(X)
        pairs = np.ascontiguousarray(pair_indices, dtype=np.int64)
        if pairs.ndim != 2 or pairs.shape[1] != 2:
            raise ValueError('pair_indices must be of shape = (n_pairs, 2)')
        return getattr(self.lib, "oracle_sumdist_" + kind)(
            C.c_void_p(X.ctypes.data), metric.encode(), C.c_int64(X.shape[0]), C.c_int64(X.shape[1]),
            _ptr(pairs, _i64p), C.c_int64(pairs.shape[0]))

    def kcenters_fit(self, X, n_clusters, metric, seed_index):
        """C restatement of _KCenters.fit (cluster/kcenters.py:79-102)."""
        X = np.ascontiguousarray(X)
        if X.dtype not in (np.float32, np.float64):
            X = X.astype(np.float64)
        kind = "f32" if X.dtype == np.float32 else "f64"
        n, m = X.shape
        ids = np.zeros(n_clusters, dtype=np.int64)
        labels = np.zeros(n, dtype=np.int64)
        distances = np.zeros(n, dtype=np.float64)
        rc = getattr(self.lib, "oracle_kcenters_fit_" + kind)(
            C.c_void_p(X.ctypes.data), C.c_int64(n), C.c_int64(m), C.c_int64(n_clusters),
            metric.encode(), C.c_int64(seed_index), _ptr(ids, _i64p), _ptr(labels, _i64p),
            _ptr(distances, _f64p))
        if rc != 0:
            raise ValueError("unknown metric %r" % metric)
        return ids, labels, distances


class Ref:
    """The reference's own libdistance headers, compiled (oracle/_ref)."""

    def __init__(self):
        build()
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        self.lib = C.CDLL(REF_SO)
        self.lib.ref_assign_nearest_double.restype = C.c_double
        self.lib.ref_assign_nearest_float.restype = C.c_double
        self.lib.ref_sumdist_double.restype = C.c_double
        self.lib.ref_sumdist_float.restype = C.c_double

    @staticmethod
    def available() -> bool:
        try:
            build()
        except Exception:
            pass
        return os.path.exists(REF_SO)

    def assign_nearest(self, X, Y, metric, X_indices=None):
        kind = _check(X, Y, metric)
        X = np.ascontiguousarray(X)
        Y = np.ascontiguousarray(Y)
        idx = None if X_indices is None else np.ascontiguousarray(X_indices, dtype=np.int64)
        n = X.shape[0] if idx is None else idx.shape[0]
        out = np.zeros(n, dtype=np.intp)
        fn = self.lib.ref_assign_nearest_double if kind == "f64" else self.lib.ref_assign_nearest_float
        inertia = fn(C.c_void_p(X.ctypes.data), C.c_void_p(Y.ctypes.data), metric.encode(),
                     _ptr(idx, _i64p), C.c_int64(X.shape[0]), C.c_int64(Y.shape[0]),
                     C.c_int64(X.shape[1]), C.c_int64(n), _ptr(out, _i64p))
        return out, inertia

    def cdist(self, XA, XB, metric):
        kind = _check(XA, XB, metric)
        XA = np.ascontiguousarray(XA)
        XB = np.ascontiguousarray(XB)
        out = np.zeros((XA.shape[0], XB.shape[0]), dtype=np.float64)
        fn = self.lib.ref_cdist_double if kind == "f64" else self.lib.ref_cdist_float
        fn(C.c_void_p(XA.ctypes.data), C.c_void_p(XB.ctypes.data), metric.encode(),
           C.c_int64(XA.shape[0]), C.c_int64(XB.shape[0]), C.c_int64(XA.shape[1]), _ptr(out, _f64p))
        return out

    def pdist(self, X, metric, X_indices=None):
        kind = _check(X, X, metric)
        X = np.ascontiguousarray(X)
        sfx = "double" if kind == "f64" else "float"
        if X_indices is None:
            n = X.shape[0]
            out = np.zeros(n * (n - 1) // 2, dtype=np.float64)
            getattr(self.lib, "ref_pdist_" + sfx)(C.c_void_p(X.ctypes.data), metric.encode(), C.c_int64(n),
                                                  C.c_int64(X.shape[1]), _ptr(out, _f64p))
        else:
            idx = np.ascontiguousarray(X_indices, dtype=np.int64)
            n = idx.shape[0]
            out = np.zeros(n * (n - 1) // 2, dtype=np.float64)
            getattr(self.lib, "ref_pdist_%s_X_indices" % sfx)(
                C.c_void_p(X.ctypes.data), metric.encode(), C.c_int64(X.shape[0]), C.c_int64(X.shape[1]),
                _ptr(idx, _i64p), C.c_int64(n), _ptr(out, _f64p))
        return out

    def sumdist(self, X, metric, pair_indices):
        kind = _check(X, X, metric)
        X = np.ascontiguousarray(X)
        pairs = np.ascontiguousarray(pair_indices, dtype=np.int64)
        fn = self.lib.ref_sumdist_double if kind == "f64" else self.lib.ref_sumdist_float
        return fn(C.c_void_p(X.ctypes.data), metric.encode(), C.c_int64(X.shape[0]), C.c_int64(X.shape[1]),
                  _ptr(pairs, _i64p), C.c_int64(pairs.shape[0]))

    def dist(self, X, y, metric, X_indices=None):
        kind = _check(X, y, metric)
        X = np.ascontiguousarray(X)
        y = np.ascontiguousarray(y)
        sfx = "double" if kind == "f64" else "float"
        if X_indices is None:
            out = np.zeros(X.shape[0], dtype=np.float64)
            getattr(self.lib, "ref_dist_" + sfx)(
                C.c_void_p(X.ctypes.data), C.c_void_p(y.ctypes.data), metric.encode(),
                C.c_int64(X.shape[0]), C.c_int64(X.shape[1]), _ptr(out, _f64p))
        else:
            idx = np.ascontiguousarray(X_indices, dtype=np.int64)
            out = np.zeros(idx.shape[0], dtype=np.float64)
            getattr(self.lib, "ref_dist_%s_X_indices" % sfx)(
                C.c_void_p(X.ctypes.data), C.c_void_p(y.ctypes.data), metric.encode(),
                C.c_int64(X.shape[0]), C.c_int64(X.shape[1]), _ptr(idx, _i64p),
                C.c_int64(idx.shape[0]), _ptr(out, _f64p))
        return out
