"""Drop-in for ``msmbuilder.libdistance`` (vector metrics) on MI355X.

Same python-level signatures, dtype rules and error types as
/root/reference/msmbuilder/libdistance/libdistance.pyx:82-270 (``assign_nearest``,
``cdist``, ``dist``, ``pdist``, ``sumdist``); the arithmetic runs in libmsmhip's exact HIP kernels
(msmbuilder_amd/csrc/distance.hip) and is bit-identical to the reference's
scalar loops.  Extension: X / X_indices may be torch CUDA tensors, in which case
the per-row outputs come back as CUDA tensors too (nothing crosses PCIe).
The ``rmsd`` metric (mdtraj's libtheobald) is out of scope and raises ValueError.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Arr, check, empty_like_placement, is_device_array

__all__ = ['assign_nearest', 'cdist', 'dist', 'pdist', 'sumdist']

VECTOR_METRICS = ("euclidean", "sqeuclidean", "cityblock", "chebyshev",
                  "canberra", "braycurtis", "hamming", "jaccard",
                  "cityblock")


def _is_array(x):
    return isinstance(x, np.ndarray) or is_device_array(x)


def _dtype_of(x):
    return np.dtype(str(x.dtype).replace("torch.", ""))


def _metric(metric):
    if isinstance(metric, bytes):
        metric = metric.decode()
    if metric not in VECTOR_METRICS:
        raise ValueError('metric must be one of %s' %
                         ', '.join("'%s'" % s for s in VECTOR_METRICS))
    return metric.encode()


def _pair_kind(X, Y, what):
    dx, dy = _dtype_of(X), _dtype_of(Y)
    if dx == np.float64 and dy == np.float64:
        return "f64", np.float64
    if dx == np.float32 and dy == np.float32:
        return "f32", np.float32
    raise TypeError(what)


def _host(y, dt):
    """per-centre arrays are host memory in the C ABI"""
    if is_device_array(y):
        y = y.detach().cpu().numpy()
    return np.ascontiguousarray(y, dtype=dt)


def _require_c_contiguous(*arrays):
    # the reference's typed memoryviews (float[:, ::1]) raise ValueError otherwise
    for a in arrays:
        if isinstance(a, np.ndarray) and not a.flags.c_contiguous:
            raise ValueError("ndarray is not C-contiguous")
        if is_device_array(a) and not a.is_contiguous():
            raise ValueError("tensor is not contiguous")


def assign_nearest(X, Y, metric, X_indices=None):
    """For each point in X (or X[X_indices]) the index of the nearest row of Y, and the
    summed distance.  libdistance.pyx:82-131 -> assign.hpp:6-91."""
    if not _is_array(X) and _is_array(Y):
        raise TypeError()
    m = _metric(metric)
    kind, dt = _pair_kind(X, Y, 'X and y must be both float32 or float64')
    _require_c_contiguous(X, Y)
    if X.ndim != 2 or Y.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2)")
    assert X.shape[1] == Y.shape[1]
    ax = Arr(X, dt)
    ay = _host(Y, dt)
    idx = None
    n = ax.shape[0]
    if X_indices is not None:
        idx = Arr(X_indices, np.int64)
        if idx.on_device != ax.on_device:
            raise ValueError("X and X_indices must live on the same side (host or device)")
        n = idx.shape[0]
    out = empty_like_placement(ax, (n,), np.intp)
    if n == 0:
        return out, 0.0   # nothing to assign (assign.hpp's loop does not run; an empty device tensor has no address)
    aout = Arr(out, np.int64)
    inertia = C.c_double(0.0)
    fn = getattr(_lib.lib(), "msm_assign_nearest_" + kind)
    check(fn(ax.vp, C.c_void_p(ay.ctypes.data), m, idx.vp if idx is not None else None,
             ax.shape[0], ay.shape[0], ax.shape[1], n, aout.vp, None, C.byref(inertia), ax.on_device))
    return out, float(inertia.value)


def cdist(XA, XB, metric):
    """Distance between each pair of the two collections: libdistance.pyx:134-179 ->
    cdist.hpp:4-49.  float64 [na, nb]."""
    if not (_is_array(XA) and _is_array(XB)):
        raise TypeError('XA and XB must be numpy arrays')
    m = _metric(metric)
    kind, dt = _pair_kind(XA, XB, 'XA and XB must be identically float32 or float64')
    _require_c_contiguous(XA, XB)
    if XA.shape[1] != XB.shape[1]:
        raise ValueError('XA and XB must have the same number of columns')
    ax = Arr(XA, dt)
    ab = _host(XB, dt)
    out = empty_like_placement(ax, (ax.shape[0], ab.shape[0]), np.float64)
    aout = Arr(out, np.float64)
    fn = getattr(_lib.lib(), "msm_cdist_" + kind)
    check(fn(ax.vp, C.c_void_p(ab.ctypes.data), m, ax.shape[0], ab.shape[0], ax.shape[1], aout.vp,
             ax.on_device))
    return out


def dist(X, y, metric, X_indices=None):
    """Distance from one point to many: libdistance.pyx:229-270 -> dist.hpp:4-80."""
    if not _is_array(X) and _is_array(y):
        raise TypeError()
    m = _metric(metric)
    kind, dt = _pair_kind(X, y, 'X and y must be both float32 or float64')
    _require_c_contiguous(X, y)
    if y.ndim != 1:
        raise ValueError("Buffer has wrong number of dimensions (expected 1, got %d)" % y.ndim)
    assert X.shape[1] == y.shape[0]
    ax = Arr(X, dt)
    ay = _host(y, dt)
    idx = None
    n = ax.shape[0]
    if X_indices is not None:
        idx = Arr(X_indices, np.int64)
        if idx.on_device != ax.on_device:
            raise ValueError("X and X_indices must live on the same side (host or device)")
        n = idx.shape[0]
    out = empty_like_placement(ax, (n,), np.float64)
    aout = Arr(out, np.float64)
    fn = getattr(_lib.lib(), "msm_dist_" + kind)
    check(fn(ax.vp, C.c_void_p(ay.ctypes.data), m, ax.shape[0], ax.shape[1],
             idx.vp if idx is not None else None, n, aout.vp, ax.on_device))
    return out


def pdist(X, metric, X_indices=None):
    """Condensed pairwise distances between the rows of X (or X[X_indices]):
    libdistance.pyx:182-226 -> pdist.hpp:4-88.  float64 [n(n-1)/2]."""
    if not _is_array(X):
        raise TypeError()
    m = _metric(metric)
    dx = _dtype_of(X)
    if dx == np.float64:
        kind, dt = "f64", np.float64
    elif dx == np.float32:
        kind, dt = "f32", np.float32
    else:
        raise TypeError('X must be float32 or float64')
    _require_c_contiguous(X)
    ax = Arr(X, dt)
    idx = None
    n = ax.shape[0]
    if X_indices is not None:
        idx = Arr(X_indices, np.int64)
        if idx.on_device != ax.on_device:
            raise ValueError("X and X_indices must live on the same side (host or device)")
        n = idx.shape[0]
    out = empty_like_placement(ax, (n * (n - 1) // 2,), np.float64)
    aout = Arr(out, np.float64)
    fn = getattr(_lib.lib(), "msm_pdist_" + kind)
    check(fn(ax.vp, m, ax.shape[0], ax.shape[1], idx.vp if idx is not None else None, n, aout.vp, ax.on_device))
    return out


def sumdist(X, metric, pair_indices):
    """Sum of the distances between the listed pairs of rows: libdistance.pyx:273-310 ->
    sumdist.hpp:4-44 (the reference adds sequentially; here an fp64 tree sum)."""
    m = _metric(metric)
    dx = _dtype_of(X)
    if dx == np.float64:
        kind, dt = "f64", np.float64
    elif dx == np.float32:
        kind, dt = "f32", np.float32
    else:
        raise TypeError('X must be both float32 or float64')
    _require_c_contiguous(X)
    ax = Arr(X, dt)
    pairs = Arr(pair_indices, np.int64)
    if len(pairs.shape) != 2 or pairs.shape[1] != 2:
        raise ValueError('pair_indices must be of shape = (n_pairs, 2)')
    if pairs.on_device != ax.on_device:
        raise ValueError("X and pair_indices must live on the same side (host or device)")
    s = C.c_double(0.0)
    fn = getattr(_lib.lib(), "msm_sumdist_" + kind)
    check(fn(ax.vp, m, ax.shape[0], ax.shape[1], pairs.vp, pairs.shape[0], C.byref(s), ax.on_device))
    return float(s.value)
