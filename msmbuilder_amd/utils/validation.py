"""Input validation with the reference's observable behaviour (msmbuilder/utils/validation.py:26-74), minus mdtraj:
a trajectory is a 2-D numpy array or a 2-D torch CUDA tensor (device-resident).

    check_iter_of_sequences  -> ValueError('sequences must be a list of sequences') unless every entry is ndim-D
    array2d                  -> at-least-2-D array; for HOST float data optionally the NaN / infinity check that
                                raises ValueError("Input contains NaN, infinity or a value too large for dtype(...)")

Device tensors are never scanned here: the kernels that read them carry the finite check (and raise the same error).
"""
import numpy as np

from .._lib import is_device_array

__all__ = ['check_iter_of_sequences', 'array2d']

_NOT_A_LIST = 'sequences must be a list of sequences'


def _is_sequence_entry(X, ndim):
    array_like = isinstance(X, np.ndarray) or is_device_array(X) or hasattr(X, 'ndim')
    return array_like and X.ndim == ndim


def check_iter_of_sequences(sequences, allow_trajectory=False, ndim=2, max_iter=None):
    """Every inspected entry must be an ``ndim``-dimensional array; inspection stops after entry ``max_iter``."""
    for position, X in enumerate(sequences):
        if not _is_sequence_entry(X, ndim):
            raise ValueError(_NOT_A_LIST)
        if max_iter is not None and position >= max_iter:
            return


def _assert_all_finite(X):
    """Host float arrays only.  The cheap test first: a finite sum proves every element finite; only when the sum is
    not finite (NaN / inf present, or an overflowing sum of finite values) is the element-wise test needed."""
    X = np.asanyarray(X)
    if X.dtype.kind not in 'fc':
        return
    if np.isfinite(X.sum()) or np.isfinite(X).all():
        return
    raise ValueError("Input contains NaN, infinity or a value too large for %r." % X.dtype)


def array2d(X, dtype=None, order=None, copy=False, force_all_finite=True):
    """At-least-2-D view / conversion of ``X``; host arrays are finite-checked unless told otherwise."""
    if is_device_array(X):
        return X.reshape(1, -1) if X.dim() < 2 else X
    out = np.asarray(np.atleast_2d(X), dtype=dtype, order=order)
    if force_all_finite:
        _assert_all_finite(out)
    if copy and out is X:
        out = np.copy(out, order='K')
    return out
