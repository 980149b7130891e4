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

__all__ = ['StandardScaler', 'MinMaxScaler', 'MaxAbsScaler', 'RobustScaler', 'column_statistics',
           'column_order_statistics', 'fold_into_tica']


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


def column_order_statistics(sequences, ranks):
    """Exact order statistics per column without sorting: ``ranks`` is an int64 array [R, F];
    the result [R, F] (dtype of the data) holds, for every column f, the element that would sit at
    index ``ranks[r, f]`` of the sorted non-NaN values of that column (NaN where the rank is negative).

    Most-significant-digit radix select on the device (``msm_col_digit_hist``): the data are
    streamed once per 11-bit digit of their order-preserving keys -- 3 passes for float32, 6 for
    float64 -- and between passes the host only walks the [R, F, 2048] histograms."""
    import torch
    arrs = [_prepare(X) for X in sequences]
    wide = any(str(a.dtype).endswith("64") for a in arrs)
    tdt, ndt, nbits = (torch.float64, np.float64, 64) if wide else (torch.float32, np.float32, 32)
    dev = [a.to(tdt) if is_device_array(a) else torch.from_numpy(np.ascontiguousarray(a, dtype=ndt)).cuda() for a in arrs]
    dev = [a.contiguous() for a in dev]
    F = dev[0].shape[1]
    ranks = np.ascontiguousarray(ranks, dtype=np.int64)
    R = ranks.shape[0]
    Arr(dev[0])
    n = len(dev)
    ptrs = (C.c_void_p * n)(*[a.data_ptr() for a in dev])
    rows = (C.c_int64 * n)(*[a.shape[0] for a in dev])
    L = _lib.lib()
    rem = np.where(ranks >= 0, ranks, 0).astype(np.int64)
    prefix = np.zeros((R, F), dtype=np.uint64)
    digits = []                      # (shift, bits): 11-bit digits from the top (the first pass counts in LDS)
    top = nbits
    while top > 0:
        b = min(11, top)
        digits.append((top - b, b))
        top -= b
    for i, (shift, bits) in enumerate(digits):
        first = i == 0
        if first:
            hist = np.zeros((1, F, 1 << bits), dtype=np.int64)
            check(L.msm_col_digit_hist(ptrs, rows, n, nbits // 8, F, F, None, R, shift, bits, hist.ctypes.data))
            hist = np.broadcast_to(hist, (R, F, 1 << bits))
        else:
            # targets of a column usually share their prefix (ranks k and k+1; nearby quantiles): count each
            # DISTINCT prefix once.  Unused slots get a prefix no key can have.
            order = np.argsort(prefix, axis=0, kind="stable")
            srt = np.take_along_axis(prefix, order, axis=0)
            newgrp = np.ones((R, F), dtype=bool)
            newgrp[1:] = srt[1:] != srt[:-1]
            gid_sorted = np.cumsum(newgrp, axis=0) - 1                     # group id of each sorted target
            U = int(gid_sorted.max()) + 1
            uniq = np.full((U, F), np.uint64(1) << np.uint64(nbits - shift - bits), dtype=np.uint64)
            np.put_along_axis(uniq, gid_sorted, srt, axis=0)
            gid = np.empty((R, F), dtype=np.int64)
            np.put_along_axis(gid, order, gid_sorted, axis=0)
            uh = np.zeros((U, F, 1 << bits), dtype=np.int64)
            check(L.msm_col_digit_hist(ptrs, rows, n, nbits // 8, F, F, uniq.ctypes.data, U, shift, bits, uh.ctypes.data))
            hist = np.take_along_axis(uh, gid[..., None], axis=0)
        cum = np.cumsum(hist, axis=-1)
        d = np.argmax(cum > rem[..., None], axis=-1)                        # digit holding the wanted rank
        below = np.where(d > 0, np.take_along_axis(cum, np.maximum(d - 1, 0)[..., None], axis=-1)[..., 0], 0)
        rem = rem - below
        prefix = (prefix << np.uint64(bits)) | d.astype(np.uint64)
    sign = np.uint64(1) << np.uint64(nbits - 1)
    full = np.uint64((1 << nbits) - 1)
    bitsu = np.where(prefix & sign, prefix & ~sign, ~prefix & full)          # undo the order-preserving map
    vals = (bitsu.astype(np.uint32).view(np.float32) if nbits == 32 else bitsu.view(np.float64)).astype(ndt)
    return np.where(ranks >= 0, vals, np.nan).astype(ndt)


def _lerp(a, b, t):
    """numpy.lib._function_base_impl._lerp (the arithmetic np.percentile's 'linear' method ends in)."""
    diff = np.subtract(b, a)
    out = np.asanyarray(np.add(a, diff * t))
    np.subtract(b, diff * (1 - t), out=out, where=t >= 0.5, casting='unsafe', dtype=type(out.dtype))
    return out


class RobustScaler(_MultiSequenceScaler):
    """Scale features using statistics that are robust to outliers: median and inter-quantile range
    (sklearn.preprocessing.RobustScaler -- the scaler of the reference's own workflow,
    tests/workflows/basic.sh; ``center_``, ``scale_``).  ``fit`` computes the exact per-column order
    statistics with ``column_order_statistics`` (one counting pass + 3 / 6 radix-select passes over the
    data) and then applies numpy's own median / percentile interpolation formulas to them."""

    def __init__(self, with_centering=True, with_scaling=True, quantile_range=(25.0, 75.0), copy=True,
                 unit_variance=False):
        self.with_centering = with_centering
        self.with_scaling = with_scaling
        self.quantile_range = quantile_range
        self.copy = copy
        self.unit_variance = unit_variance

    def partial_fit(self, sequence, y=None):
        # not an online estimator upstream either: preprocessing/base.py:140-157 maps partial_fit to fit
        return self.fit([sequence])

    def fit(self, sequences, y=None):
        check_iter_of_sequences(sequences)
        q_min, q_max = self.quantile_range
        if not 0 <= q_min <= q_max <= 100:
            raise ValueError("Invalid quantile range: %s" % str(self.quantile_range))
        sequences = list(sequences)
        st = column_statistics(sequences)                  # non-NaN counts, infinity check
        n = st["n"].astype(np.int64)
        F = len(n)
        self.n_features_in_ = F
        self._stats = np.stack([st["n"], st["mean"], st["m2"], st["min"], st["max"]])
        ranks = []
        if self.with_centering:
            ranks += [np.where(n > 0, (n - 1) // 2, -1), np.where(n > 0, n // 2, -1)]
        gammas = []
        if self.with_scaling:
            for q in (q_min, q_max):
                quant = np.true_divide(q, 100)
                virt = (n - 1) * quant                                   # numpy's 'linear' method: get_virtual_index
                prev = np.floor(virt).astype(np.int64)
                nxt = prev + 1
                above = virt >= n - 1
                prev, nxt = np.where(above, n - 1, prev), np.where(above, n - 1, nxt)
                below = virt < 0
                prev, nxt = np.where(below, 0, prev), np.where(below, 0, nxt)
                gammas.append(np.asanyarray(virt - np.floor(virt), dtype=np.float64))
                ranks += [np.where(n > 0, prev, -1), np.where(n > 0, nxt, -1)]
        vals = column_order_statistics(sequences, np.stack(ranks)) if ranks else np.zeros((0, F))
        k = 0
        if self.with_centering:
            # np.median: mean of the two middle elements in the array's own precision
            self.center_ = np.mean(np.stack([vals[0], vals[1]]), axis=0)
            k = 2
        else:
            self.center_ = None
        if self.with_scaling:
            qs = [_lerp(vals[k], vals[k + 1], gammas[0]), _lerp(vals[k + 2], vals[k + 3], gammas[1])]
            self.scale_ = _handle_zeros_in_scale(np.asarray(qs[1]) - np.asarray(qs[0]))
            if self.unit_variance:
                from scipy import stats
                self.scale_ = self.scale_ / (stats.norm.ppf(q_max / 100.0) - stats.norm.ppf(q_min / 100.0))
        else:
            self.scale_ = None
        return self

    def partial_transform(self, sequence):
        self._check_fitted(sequence)
        return _apply(sequence, self.center_ if self.with_centering else None,
                      self.scale_ if self.with_scaling else None, 0)

    def partial_inverse_transform(self, sequence):
        self._check_fitted(sequence)
        return _apply(sequence, self.center_ if self.with_centering else None,
                      self.scale_ if self.with_scaling else None, 1)

    def inverse_transform(self, sequences):
        return [self.partial_inverse_transform(X) for X in sequences]


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
