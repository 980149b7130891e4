"""``_transition_counts`` of msmbuilder.msm (/root/reference/msmbuilder/msm/core.py:487-596) with
the pair counting on the GPU.

Same signature, return value and label semantics: labels may be any orderable objects,
``mapping`` sends them to ``0 .. n_states-1`` in sorted order, ``None`` / NaN are not states and
kill the pairs they take part in, ``sliding_window=False`` strides the sequences first, and the
counts are divided by ``lag_time``.

Integer label sequences -- what the clustering kernels produce, host arrays or ``torch`` CUDA
tensors straight from ``KCenters.labels_`` -- never leave the device: class discovery is
``msm_label_range`` + ``msm_label_histogram`` and the counting ``msm_transition_counts``
(int64, exact).  Other label types (strings, floats with NaN, ``None``) are encoded to integer
codes on the host, which is label bookkeeping, and then counted by the same kernel.
"""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import Arr, check, is_device_array

_MAX_DENSE_RANGE = 1 << 26     # label spans wider than this are compacted on the host first


def _as_int64(y):
    if is_device_array(y):
        import torch
        return y.to(torch.int64).contiguous()
    return np.ascontiguousarray(y, dtype=np.int64)


def _tables(seqs):
    n = len(seqs)
    on_dev = is_device_array(seqs[0]) if n else False
    if on_dev:
        Arr(seqs[0])
        ptrs = (C.c_void_p * n)(*[a.data_ptr() for a in seqs])
    else:
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in seqs])
    rows = (C.c_int64 * n)(*[int(a.shape[0]) for a in seqs])
    return ptrs, rows, n, int(on_dev)


def _is_integer_sequence(y):
    if is_device_array(y):
        return not (y.is_floating_point() or y.is_complex()) and y.dim() == 1
    return isinstance(y, np.ndarray) and y.dtype.kind in "iu" and y.ndim == 1


def _count(seqs, lag_time, lo, remap, n_bins, n_states):
    counts = np.zeros((n_states, n_states), dtype=np.int64)
    if n_states == 0:
        return counts
    # one launch per placement class (host sequences are staged together)
    for group in ([s for s in seqs if is_device_array(s)], [s for s in seqs if not is_device_array(s)]):
        if not group:
            continue
        ptrs, rows, n, on_dev = _tables(group)
        part = np.zeros((n_states, n_states), dtype=np.int64)
        check(_lib.lib().msm_transition_counts(ptrs, rows, n, on_dev, int(lag_time), int(lo),
                                               None if remap is None else remap.ctypes.data, int(n_bins),
                                               int(n_states), part.ctypes.data))
        counts += part
    return counts


def _transition_counts(sequences, lag_time=1, sliding_window=True):
    """Count the number of directed transitions in a collection of sequences in a discrete space.

    Parameters
    ----------
    sequences : list of array-like
        List of sequences. Each sequence should be a 1D iterable of state labels. Labels can
        be integers, strings, or other orderable objects.
    lag_time : int
        The time (index) delay for the counts.
    sliding_window : bool
        When lag_time > 1, consider *all* ``N = lag_time`` strided sequences starting from index
        0, 1, 2, ..., ``lag_time - 1``. The total, raw counts will be divided by ``N``. When this
        is False, only start from index 0.

    Returns
    -------
    counts : array, shape=(n_states, n_states)
    mapping : dict
        Mapping from the items in the sequences to the indices in ``(0, n_states-1)``.
    """
    if (not sliding_window) and lag_time > 1:
        return _transition_counts([X[::lag_time] for X in sequences], lag_time=1)

    sequences = [y if is_device_array(y) else np.asarray(y) for y in sequences]
    if len(sequences) > 0 and all(_is_integer_sequence(y) for y in sequences):
        return _integer_path(sequences, lag_time)
    return _generic_path([y.cpu().numpy() if is_device_array(y) else y for y in sequences], lag_time)


def _integer_path(sequences, lag_time):
    label_dtype = None if is_device_array(sequences[0]) else sequences[0].dtype
    seqs = [_as_int64(y) for y in sequences]
    L = _lib.lib()
    lo, hi = None, None
    for group in ([s for s in seqs if is_device_array(s)], [s for s in seqs if not is_device_array(s)]):
        if not group:
            continue
        ptrs, rows, n, on_dev = _tables(group)
        a, b, tot = C.c_int64(0), C.c_int64(-1), C.c_int64(0)
        check(L.msm_label_range(ptrs, rows, n, on_dev, C.byref(a), C.byref(b), C.byref(tot)))
        if b.value >= a.value:
            lo = a.value if lo is None else min(lo, a.value)
            hi = b.value if hi is None else max(hi, b.value)
    if lo is None:                                   # no labels at all
        return np.zeros((0, 0)), {}
    n_bins = hi - lo + 1
    if n_bins > _MAX_DENSE_RANGE:                    # sparse huge labels: compact on the host
        return _generic_path([s.cpu().numpy() if is_device_array(s) else s for s in seqs], lag_time,
                             label_dtype=label_dtype)
    hist = np.zeros(n_bins, dtype=np.int64)
    for group in ([s for s in seqs if is_device_array(s)], [s for s in seqs if not is_device_array(s)]):
        if not group:
            continue
        ptrs, rows, n, on_dev = _tables(group)
        part = np.zeros(n_bins, dtype=np.int64)
        check(L.msm_label_histogram(ptrs, rows, n, on_dev, int(lo), int(n_bins), part.ctypes.data))
        hist += part
    present = np.flatnonzero(hist)
    classes = (present + lo).astype(label_dtype if label_dtype is not None else np.int64)
    n_states = len(classes)
    mapping = dict(zip(classes, range(n_states)))
    if n_states == n_bins:                            # contiguous labels: state = label - lo
        remap = None
    else:
        remap = np.full(n_bins, -1, dtype=np.int32)
        remap[present] = np.arange(n_states, dtype=np.int32)
    counts = _count(seqs, lag_time, lo, remap, n_bins, n_states).astype(float)
    counts /= float(lag_time)
    return counts, mapping


def _generic_path(sequences, lag_time, label_dtype=None):
    """Arbitrary labels: sorted unique classes on the host (core.py:544-558), integer codes with -1
    for NaN / None, counting on the device."""
    classes = np.unique(np.concatenate(sequences))
    contains_nan = (classes.dtype.kind == 'f') and np.any(np.isnan(classes))
    contains_none = any(c is None for c in classes)
    if contains_nan:
        classes = classes[~np.isnan(classes)]
    if contains_none:
        classes = [c for c in classes if c is not None]
    if label_dtype is not None:
        classes = np.asarray(classes).astype(label_dtype)
    n_states = len(classes)
    mapping = dict(zip(classes, range(n_states)))
    codes = []
    for y in sequences:
        y = np.asarray(y)
        code = np.full(len(y), -1, dtype=np.int64)
        if n_states and y.dtype.kind in "fiuUS":      # sorted classes: one vectorised lookup
            valid = ~np.isnan(y) if y.dtype.kind == "f" else np.ones(len(y), dtype=bool)
            cls = np.asarray(classes)
            code[valid] = np.searchsorted(cls, y[valid].astype(cls.dtype, copy=False))   # mixed str / int inputs: compare in the classes' dtype
        elif n_states:
            for i, v in enumerate(y):
                if v is None or (isinstance(v, (float, np.floating)) and np.isnan(v)):
                    continue
                code[i] = mapping[v]
        codes.append(code)
    counts = _count(codes, lag_time, 0, None, n_states, n_states).astype(float)
    counts /= float(lag_time)
    return counts, mapping
