"""Input validation with the reference's semantics
(/root/reference/msmbuilder/utils/validation.py:26-74), minus mdtraj: sequences
are 2-D numpy arrays or 2-D torch CUDA tensors (device-resident trajectories).
"""
import numpy as np

from .._lib import is_device_array

__all__ = ['check_iter_of_sequences', 'array2d']


def check_iter_of_sequences(sequences, allow_trajectory=False, ndim=2, max_iter=None):
    """validation.py:26-55: every checked entry must be an ndim-D array, else
    ``ValueError('sequences must be a list of sequences')``."""
    value = True
    for i, X in enumerate(sequences):
        if not (isinstance(X, np.ndarray) or is_device_array(X)) and not hasattr(X, 'ndim'):
            value = False
            break
        if X.ndim != ndim:
            value = False
            break
        if max_iter is not None and i >= max_iter:
            break
    if not value:
        raise ValueError('sequences must be a list of sequences')


def _assert_all_finite(X):
    """validation.py:68-74 (host arrays; device arrays are checked inside the kernels)."""
    X = np.asanyarray(X)
    if (X.dtype.char in np.typecodes['AllFloat'] and not np.isfinite(X.sum())
            and not np.isfinite(X).all()):
        raise ValueError("Input contains NaN, infinity"
                         " or a value too large for %r." % X.dtype)


def array2d(X, dtype=None, order=None, copy=False, force_all_finite=True):
    """validation.py:58-65: at-least-2-D array, finite-checked."""
    if is_device_array(X):
        return X if X.dim() >= 2 else X.reshape(1, -1)
    X_2d = np.asarray(np.atleast_2d(X), dtype=dtype, order=order)
    if force_all_finite:
        _assert_all_finite(X_2d)
    if X is X_2d and copy:
        X_2d = np.copy(X_2d, order='K')
    return X_2d
