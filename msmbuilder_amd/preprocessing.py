"""Column scalers in front of tICA, computed on the GPU.

Drop-ins for ``msmbuilder.preprocessing.StandardScaler / MinMaxScaler / MaxAbsScaler``
(/root/reference/msmbuilder/preprocessing/__init__.py:56-83: scikit-learn's scalers behind
the multi-sequence mixins of preprocessing/base.py:14-199): same constructor arguments,
``fit(sequences)`` / ``partial_fit(X)`` / ``transform(sequences)`` / ``partial_transform(X)`` /
``fit_transform``, same fitted attributes.  ``fit`` is ONE streaming pass of
``msm_colstats`` over the trajectories (host arrays are staged, ``torch`` CUDA tensors are read
in place); ``transform`` is one ``msm_scale_apply`` per trajectory and returns arrays of the
input's dtype and placement.

The arithmetic is scikit-learn's (third party, version unpinned by the reference; the tests
compare against the scikit-learn installed next to this package): float64 statistics with
NaN treated as a missing value, batches merged with the Chan/Golub/LeVeque update
(``sklearn.utils.extmath._incremental_mean_and_var``), near-constant columns get scale 1
(``_is_constant_feature`` / ``_handle_zeros_in_scale``), and ``transform`` rounds after every
step like numpy's in-place operators do.

``fold_into_tica`` applies a fitted StandardScaler to a tICA model ALGEBRAICALLY: the model
is accumulated on the raw trajectories and behaves as if it had been fitted on
``scaler.transform(sequences)`` -- the scaled copy of the data set is never written.
"""
import ctypes as C

import numpy as np
import sklearn.base
import sklearn.exceptions

from . import _lib
from ._lib import Arr, check, empty_like_placement, is_device_array
from .base import BaseEstimator
from .utils.validation import array2d, check_iter_of_sequences

__all__ = ['StandardScaler', 'MinMaxScaler', 'MaxAbsScaler', 'column_statistics', 'fold_into_tica']


def _prepare(X):
    X = array2d(X, force_all_finite=False)
    if is_device_array(X):
        import torch
        if X.dtype not in (torch.float32, torch.float64):
            X = X.to(torch.float64)
        return X.contiguous()
    if X.dtype not in (np.float32, np.float64):
        X = np.asarray(X, dtype=np.float64)
    return np.ascontiguousarray(X)


def column_statistics(sequences):
    """One pass over a list of (n_i, F) arrays: dict of per-column ``n`` (non-NaN count),
    ``mean``, ``m2`` (sum of squared deviations), ``min``, ``max`` -- float64 arrays of
    length F -- and ``dtype``, the common floating type of the inputs.  Raises ValueError on infinities (scikit-learn's ``ensure_all_finite='allow-nan'``)."""
    arrs = [_prepare(X) for X in sequences]
    if not arrs:
        raise ValueError("need at least one array")
    F = arrs[0].shape[1]
    for a in arrs:
        if a.shape[1] != F:
            raise ValueError("X has %d features, but %d were expected" % (a.shape[1], F))
    acc = None
    classes = {}
    for a in arrs:
        classes.setdefault((is_device_array(a), 8 if str(a.dtype).endswith("64") else 4), []).append(a)
    L = _lib.lib()
    for (on_dev, nbytes), group in classes.items():
        n = len(group)
        if on_dev:
            Arr(group[0])  # device / stream binding once; raw pointers for the rest
            ptrs = (C.c_void_p * n)(*[a.data_ptr() for a in group])
        else:
            ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in group])
        rows = (C.c_int64 * n)(*[a.shape[0] for a in group])
        out = np.empty((5, F))
        has_inf = C.c_int(0)
        check(L.msm_colstats(ptrs, rows, n, nbytes, F, F, int(on_dev), out.ctypes.data, C.byref(has_inf)))
        if has_inf.value:
            raise ValueError("Input X contains infinity or a value too large for dtype('%s')."
                             % ("float64" if nbytes == 8 else "float32"))
        acc = out if acc is None else _merge(acc, out)
    dt = np.result_type(*[np.float64 if str(a.dtype).endswith("64") else np.float32 for a in arrs])
    return dict(n=acc[0], mean=acc[1], m2=acc[2], min=acc[3], max=acc[4], dtype=dt)


def _merge(a, b):
    """Chan/Golub/LeVeque merge of two [5, F] statistics blocks (column-wise)."""
    na, nb = a[0], b[0]
    tot = na + nb
    with np.errstate(invalid="ignore", divide="ignore"):
        w = np.where(tot > 0, nb / np.where(tot > 0, tot, 1), 0.0)
    delta = b[1] - a[1]
    out = np.empty_like(a)
    out[0] = tot
    out[1] = np.where(nb > 0, a[1] + delta * w, a[1])
    out[2] = np.where(nb > 0, a[2] + b[2] + delta * delta * na * w, a[2])
    out[3] = np.minimum(a[3], b[3])
    out[4] = np.maximum(a[4], b[4])
    return out


def _apply(X, shift, scale, mode):
    X = _prepare(X)
    ax = Arr(X)
    out = empty_like_placement(ax, tuple(X.shape), np.float64 if str(X.dtype).endswith("64") else np.float32)
    ao = Arr(out)
    F = X.shape[1]
    sh = None if shift is None else np.ascontiguousarray(np.broadcast_to(shift, (F,)), dtype=np.float64)
    sc = None if scale is None else np.ascontiguousarray(np.broadcast_to(scale, (F,)), dtype=np.float64)
    check(_lib.lib().msm_scale_apply(ax.ptr, 8 if str(X.dtype).endswith("64") else 4, X.shape[0], F, F,
                                     None if sh is None else sh.ctypes.data,
                                     None if sc is None else sc.ctypes.data, mode, ao.ptr, F,
                                     int(is_device_array(X))))
    return out


def _handle_zeros_in_scale(scale, constant_mask=None, dtype=np.float64):
    scale = np.array(scale, dtype=dtype, copy=True)
    if constant_mask is None:
        constant_mask = scale < 10 * np.finfo(scale.dtype).eps
    scale[constant_mask] = 1.0
    return scale


class _MultiSequenceScaler(BaseEstimator, sklearn.base.TransformerMixin):
    """fit / transform over lists of sequences (preprocessing/base.py:31-199)."""

    def _reset(self):
        self._stats = None
        self._dtype = None

    def fit(self, sequences, y=None):
        check_iter_of_sequences(sequences)
        self._reset()
        self._absorb(column_statistics(sequences))
        return self

    def partial_fit(self, sequence, y=None):
        st = column_statistics([sequence])
        self._absorb(st)
        return self

    def _absorb(self, st):
        new = np.stack([st["n"], st["mean"], st["m2"], st["min"], st["max"]])
        prev = getattr(self, "_stats", None)
        self._stats = new if prev is None else _merge(prev, new)
        seen = getattr(self, "_dtype", None)
        self._dtype = st["dtype"] if seen is None else np.result_type(seen, st["dtype"])
        self.n_features_in_ = new.shape[1]
        self._finalise()

    def transform(self, sequences):
        check_iter_of_sequences(sequences)
        return [self.partial_transform(X) for X in sequences]

    def fit_transform(self, sequences, y=None):
        return self.fit(sequences).transform(sequences)

    def _check_fitted(self, X):
        if getattr(self, "_stats", None) is None:
            raise sklearn.exceptions.NotFittedError(
                "This %s instance is not fitted yet. Call 'fit' with appropriate arguments before using this "
                "estimator." % type(self).__name__)
        if np.shape(X)[1] != self.n_features_in_:
            raise ValueError("X has %d features, but %s is expecting %d features as input."
                             % (np.shape(X)[1], type(self).__name__, self.n_features_in_))

    def _n_samples_seen(self):
        n = self._stats[0].astype(np.int64)
        return int(n[0]) if np.ptp(n) == 0 else n


class StandardScaler(_MultiSequenceScaler):
    """Standardize features by removing the mean and scaling to unit variance
    (sklearn.preprocessing.StandardScaler; ``mean_``, ``var_``, ``scale_``, ``n_samples_seen_``)."""

    def __init__(self, copy=True, with_mean=True, with_std=True):
        self.copy = copy
        self.with_mean = with_mean
        self.with_std = with_std

    def _finalise(self):
        n, mean, m2 = self._stats[0], self._stats[1], self._stats[2]
        self.n_samples_seen_ = self._n_samples_seen()
        if not self.with_mean and not self.with_std:
            self.mean_, self.var_, self.scale_ = None, None, None
            return
        self.mean_ = mean.copy()
        if self.with_std:
            with np.errstate(invalid="ignore", divide="ignore"):
                self.var_ = m2 / n
            eps = np.finfo(np.float64).eps
            upper = n * eps * self.var_ + (n * self.mean_ * eps) ** 2      # sklearn _is_constant_feature
            self.scale_ = _handle_zeros_in_scale(np.sqrt(self.var_), constant_mask=self.var_ <= upper)
        else:
            self.var_, self.scale_ = None, None

    def partial_transform(self, sequence):
        self._check_fitted(sequence)
        return _apply(sequence, self.mean_ if self.with_mean else None, self.scale_ if self.with_std else None, 0)

    def inverse_transform(self, sequences):
        return [self.partial_inverse_transform(X) for X in sequences]

    def partial_inverse_transform(self, sequence):
        self._check_fitted(sequence)
        # numpy: X *= scale_; X += mean_
        return _apply(sequence, self.mean_ if self.with_mean else None, self.scale_ if self.with_std else None, 1)


class MinMaxScaler(_MultiSequenceScaler):
    """Transform features by scaling each feature to a given range
    (sklearn.preprocessing.MinMaxScaler; ``min_``, ``scale_``, ``data_min_``, ``data_max_``, ``data_range_``)."""

    def __init__(self, feature_range=(0, 1), copy=True, clip=False):
        self.feature_range = feature_range
        self.copy = copy
        self.clip = clip

    def _finalise(self):
        lo, hi = self.feature_range
        if lo >= hi:
            raise ValueError("Minimum of desired feature range must be smaller than maximum. Got %s."
                             % str(self.feature_range))
        self.n_samples_seen_ = int(self._stats[0].max())
        # scikit-learn keeps these in the dtype of the data (float32 in, float32 parameters)
        self.data_min_ = self._stats[3].astype(self._dtype)
        self.data_max_ = self._stats[4].astype(self._dtype)
        self.data_range_ = self.data_max_ - self.data_min_
        self.scale_ = (hi - lo) / _handle_zeros_in_scale(self.data_range_, dtype=self._dtype)
        self.min_ = lo - self.data_min_ * self.scale_

    def partial_transform(self, sequence):
        self._check_fitted(sequence)
        out = _apply(sequence, self.min_, self.scale_, 1)
        if self.clip:
            lo, hi = self.feature_range
            out = out.clamp(lo, hi) if is_device_array(out) else np.clip(out, lo, hi, out=out)
        return out


class MaxAbsScaler(_MultiSequenceScaler):
    """Scale each feature by its maximum absolute value
    (sklearn.preprocessing.MaxAbsScaler; ``scale_``, ``max_abs_``, ``n_samples_seen_``)."""

    def __init__(self, copy=True):
        self.copy = copy

    def _finalise(self):
        self.n_samples_seen_ = int(self._stats[0].max())
        self.max_abs_ = np.maximum(np.abs(self._stats[3]), np.abs(self._stats[4])).astype(self._dtype)
        self.scale_ = _handle_zeros_in_scale(self.max_abs_, dtype=self._dtype)

    def partial_transform(self, sequence):
        self._check_fitted(sequence)
        return _apply(sequence, None, self.scale_, 0)


def fold_into_tica(scaler, tica, sequences=None):
    """Make ``tica`` behave as if it had been fitted on ``scaler.transform(sequences)``.

    With x' = (x - m) / s the centred moments transform as mu' = (mu - m) / s,
    OC' = D OC D and S' = D S D (D = diag(1/s)): the tICA accumulators stay those of the RAW
    data (one pass of the covariance kernel, no scaled copy of the data set) and the
    scaling is applied to the F x F moments on the host; ``tica.transform`` then takes RAW
    trajectories too, with D folded into the projection matrix.  If ``sequences`` is given
    both models are fitted first (one column scan + one covariance pass).  Returns ``tica``.
    """
    if sequences is not None:
        scaler.fit(sequences)
        tica.fit(sequences)
    if not isinstance(scaler, StandardScaler):
        raise TypeError("fold_into_tica needs a StandardScaler (an affine per-column map with known shift/scale)")
    F = scaler.n_features_in_
    shift = scaler.mean_ if scaler.with_mean else np.zeros(F)
    scale = scaler.scale_ if scaler.with_std else np.ones(F)
    tica.set_input_scaling(shift, scale)
    return tica
