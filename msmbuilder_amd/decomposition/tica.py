"""tICA on MI355X: drop-in for ``msmbuilder.decomposition.tICA``.

Same constructor, methods, learned attributes, warnings and exceptions as
/root/reference/msmbuilder/decomposition/tica.py:26-489.  What differs is where
the work happens: ``_fit`` (tica.py:401-424 -- the float64 up-cast and the three
dgemm per trajectory) is replaced by libmsmhip's MFMA accumulation kernel
(msmbuilder_amd/csrc/tica.hip) and the accumulators live in HBM; ``transform``
(tica.py:329-352) is the fused projection kernel.  The O(F^2)/O(F^3)
finalisation (means, shrinkage, generalized eigensolve: tica.py:167-259,492-524)
is unchanged numpy/scipy on the host, fed by one export of the accumulators.

Inputs may be numpy arrays (host; staged over PCIe) or 2-D torch CUDA tensors
(device-resident: nothing crosses PCIe, ``transform`` returns CUDA tensors).

Precision: ``MSMBUILDER_AMD_TICA_MODE=f32`` (default) accumulates float32 inputs
with exact-fp32 MFMA in <=4096-frame chunks merged in fp64 (eigenvalues agree
with the float64 reference to rtol 1e-5, typically 1e-7); ``f64`` uses the fp64
MFMA on widened inputs and reproduces the reference's float64 arithmetic up to
summation order (rtol 1e-10); ``bf16`` rounds the inputs to bfloat16 for the 16x faster bf16 MFMA
(rtol 1e-3); ``bf16x2`` splits every value into two bfloat16 terms and forms all four products
(fp32-class accuracy, rtol 1e-5, at twice the fp32-MFMA speed).  float64 inputs always take the fp64
kernel.

Multi-GPU: every rank fits its own shard of trajectories, then
``allreduce()`` sums the packed accumulators with one RCCL all-reduce
(torch.distributed); see msmbuilder_amd/parallel.py.
"""
from __future__ import print_function, division, absolute_import

import ctypes as C
import os
import warnings

import numpy as np
from sklearn.base import TransformerMixin

from .. import _lib
from .._lib import Arr, check, is_device_array
from ..base import BaseEstimator
from ..utils import check_iter_of_sequences, array2d
from ..utils.validation import _assert_all_finite
from . import _moments
from ._moments import rao_blackwell_ledoit_wolf  # noqa: F401  (public name in the reference module)

__all__ = ['tICA']

_BATCH_BYTES = 1 << 30  # host trajectories are shipped in groups of about this size

# Released device handles are parked here and re-used (after msm_tica_reset) by the next model of the
# same shape: creating one costs eight hipMalloc and destroying one eight hipFree, each of which
# synchronises the device -- tens of ms when models are fitted in a loop.
_HANDLE_POOL = {}
_HANDLE_POOL_MAX = 2


def _acquire_handle(n_features, lag_time, mode, sym=""):
    key = (int(n_features), int(lag_time), int(mode), str(sym))
    free = _HANDLE_POOL.get(key)
    if free:
        h = free.pop()
        check(_lib.lib().msm_tica_reset(h))
        return h
    h = C.c_void_p()
    check(_lib.lib().msm_tica_create(C.byref(h), key[0], key[1], key[2]))
    return h


def _park_handle(h, n_features, lag_time, mode, sym=""):
    key = (int(n_features), int(lag_time), int(mode), str(sym))
    free = _HANDLE_POOL.setdefault(key, [])
    if len(free) < _HANDLE_POOL_MAX:
        free.append(h)
    else:
        _lib.lib().msm_tica_destroy(h)


def release_parked_handles():
    """Destroy the device handles parked for re-use (each keeps its accumulator slabs and, in the bf16 modes, its packed
    image: tens of GB at configs[4]'s size).  Models in use are not affected."""
    for free in _HANDLE_POOL.values():
        while free:
            _lib.lib().msm_tica_destroy(free.pop())


def _mode_from_env():
    m = os.environ.get("MSMBUILDER_AMD_TICA_MODE", "f32").lower()
    modes = {"f32": _lib.TICA_F32, "f64": _lib.TICA_F64, "bf16": _lib.TICA_BF16, "bf16x2": _lib.TICA_BF16X2}
    if m not in modes:
        raise ValueError("MSMBUILDER_AMD_TICA_MODE must be one of %s" % sorted(modes))
    return modes[m]


class tICA(BaseEstimator, TransformerMixin):
    """tICA: the linear combinations of the input features that decorrelate most slowly.

    The estimator accumulates, over all pairs of frames ``lag_time`` apart, the symmetrised time-lagged
    second moment and the instantaneous covariance of the features, and solves the generalized symmetric
    eigenproblem between the two.  The eigenvectors with the largest eigenvalues (autocorrelations at the lag)
    are the slow coordinates; ``transform`` projects mean-free frames onto the first ``n_components`` of them.
    Constructor arguments, fitted attributes, warnings and exceptions are those of the reference class
    (/root/reference/msmbuilder/decomposition/tica.py:26-148); the work runs in libmsmhip.

    Parameters
    ----------
    n_components : int or None
        How many slow coordinates to keep; ``None`` keeps as many as there are features.
    lag_time : int
        Offset, in frames, between the two members of a pair (x_t, x_{t+lag_time}).
    shrinkage : float in [0, 1] or None
        Weight of the scaled identity mixed into the covariance estimate.  ``None`` (default) computes it from
        the data with the Rao-Blackwellised Ledoit-Wolf formula.
    kinetic_mapping : bool
        Multiply every projected coordinate by its eigenvalue, so Euclidean distances approximate kinetic ones.
    commute_mapping : bool
        Multiply every projected coordinate by sqrt(t_reg / 2), t_reg a regularised implied timescale (commute
        distances).  Mutually exclusive with ``kinetic_mapping``.

    Attributes
    ----------
    components_ : (n_components, n_features)   rows are the slow directions
    eigenvalues_, eigenvectors_, timescales_   top ``n_components`` solutions, eigenvalues descending
    means_, covariance_, offset_correlation_   the centred moments the eigenproblem is built from
    shrinkage_                                 the intensity actually used
    n_observations_, n_sequences_              frames and trajectories seen so far
    score_                                     sum of the kept eigenvalues
    """

    def __init__(self, n_components=None, lag_time=1, shrinkage=None,
                 kinetic_mapping=False, commute_mapping=False):
        self.n_components = n_components
        self.lag_time = lag_time
        self.shrinkage = shrinkage
        self.shrinkage_ = None
        self.kinetic_mapping = kinetic_mapping
        self.commute_mapping = commute_mapping
        if self.kinetic_mapping and self.commute_mapping:
            raise ValueError("Can't have both kinetic mapping and "
                             "commute mapping. Please only use one.")
        self.n_features = None
        self.n_observations_ = None
        self.n_sequences_ = None

        self._initialized = False

        # device side: opaque msm_tica_t* holding the accumulators in HBM
        self._handle = None
        # host mirrors, refreshed from the device when stale
        # X[:-lag].T @ X[lag:]
        self._outer_0_to_T_lagged = None
        # X[:-lag].sum(0) and X[lag:].sum(0)
        self._sum_0_to_TminusTau = None
        self._sum_tau_to_T = None
        # X[:-lag].T @ X[:-lag] + X[lag:].T @ X[lag:]  (the reference keeps the two terms
        # apart but only ever reads their sum, tica.py:245)
        self._outer_gram_sum = None
        self._host_stale = False
        # optional per-column affine map of the INPUT, x' = (x - shift) / scale, applied
        # algebraically to the F x F moments (set_input_scaling / preprocessing.fold_into_tica)
        self._input_shift = None
        self._input_scale = None

        # Cached results of the eigendecompsition
        self._components_ = None
        self._eigenvectors_ = None
        self._eigenvalues_ = None

        # are our current tICs dirty? set by _fit
        self._is_dirty = True

    # ------------------------------------------------------------------ device state
    def _initialize(self, n_features):
        if self._initialized:
            return
        if self.n_components is None:
            self.n_components = n_features
        self.n_features = n_features
        self.n_observations_ = 0
        self.n_sequences_ = 0
        self._release()
        _lib.ensure_device()
        self._handle_key = (int(n_features), int(self.lag_time), _mode_from_env(), os.environ.get("MSM_TICA_SYM", "") + "/" + os.environ.get("MSM_TICA_SYMW", ""))
        self._handle = _acquire_handle(*self._handle_key)
        self._outer_0_to_T_lagged = np.zeros((n_features, n_features))
        self._sum_0_to_TminusTau = np.zeros(n_features)
        self._sum_tau_to_T = np.zeros(n_features)
        self._outer_gram_sum = np.zeros((n_features, n_features))
        self._host_stale = False
        self._mu_raw = None
        self._initialized = True

    def _release(self):
        h = getattr(self, "_handle", None)
        if h is not None:
            try:
                _park_handle(h, *self._handle_key)
            except Exception:
                pass
        self._handle = None

    def __del__(self):
        self._release()

    def _ensure_handle(self):
        """(Re)create the device handle from the host mirrors (after unpickling)."""
        if self._handle is None and self._initialized:
            _lib.ensure_device()
            self._handle_key = (int(self.n_features), int(self.lag_time), _mode_from_env(), os.environ.get("MSM_TICA_SYM", "") + "/" + os.environ.get("MSM_TICA_SYMW", ""))
            h = self._handle = _acquire_handle(*self._handle_key)
            c = np.ascontiguousarray(self._outer_0_to_T_lagged, dtype=np.float64)
            g = np.ascontiguousarray(self._outer_gram_sum, dtype=np.float64)
            s0 = np.ascontiguousarray(self._sum_0_to_TminusTau, dtype=np.float64)
            st = np.ascontiguousarray(self._sum_tau_to_T, dtype=np.float64)
            check(_lib.lib().msm_tica_import(h, c.ctypes.data, g.ctypes.data, s0.ctypes.data,
                                             st.ctypes.data, int(self.n_observations_),
                                             int(self.n_sequences_)))

    @property
    def _lagged_symmetrised(self):
        """True when ``_outer_0_to_T_lagged`` holds (C + C^T)/2 instead of the raw X[:-tau].T @ X[tau:]
        (fp32 symmetric sum/difference kernel; only the symmetric part is ever read, tica.py:234-241)."""
        if self._handle is None:
            return False
        flag = C.c_int(0)
        check(_lib.lib().msm_tica_lagged_symmetrised(self._handle, C.byref(flag)))
        return bool(flag.value)

    def _pull(self):
        """Refresh the host mirrors of the accumulators (one D2H of 2F^2+2F doubles)."""
        if not self._host_stale:
            return
        F = self.n_features
        c = np.empty((F, F))
        g = np.empty((F, F))
        s0 = np.empty(F)
        st = np.empty(F)
        nobs, nseq = C.c_int64(0), C.c_int64(0)
        check(_lib.lib().msm_tica_export(self._handle, c.ctypes.data, g.ctypes.data, s0.ctypes.data,
                                         st.ctypes.data, C.byref(nobs), C.byref(nseq)))
        self._outer_0_to_T_lagged, self._outer_gram_sum = c, g
        self._sum_0_to_TminusTau, self._sum_tau_to_T = s0, st
        self._host_stale = False

    def __getstate__(self):
        if self._initialized and self._handle is not None:
            self._pull()
        d = dict(self.__dict__)
        d["_handle"] = None
        d.pop("_handle_key", None)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self._handle = None

    @classmethod
    def from_reference_state(cls, state, **subclass_kwargs):
        """A model carrying on from a fitted REFERENCE tICA: ``state`` is that object's ``__dict__`` (or any mapping
        with the same keys) -- hyper-parameters ``n_components, lag_time, shrinkage, kinetic_mapping,
        commute_mapping``, counters ``n_features, n_observations_, n_sequences_`` and the accumulators the reference
        declares (tica.py:128-148): ``_outer_0_to_T_lagged, _sum_0_to_TminusTau, _sum_tau_to_T,
        _outer_0_to_TminusTau, _outer_offset_to_T``.  The moments go to the device through ``msm_tica_import`` (the two
        Gram halves as their sum, the only form the reference ever reads: tica.py:245); ``partial_fit``, ``transform``
        and every fitted attribute then behave as if this class had seen the reference's data."""
        get = state.get if hasattr(state, "get") else (lambda k, d=None: getattr(state, k, d))
        m = cls(n_components=get("n_components"), lag_time=get("lag_time", 1), shrinkage=get("shrinkage"),
                kinetic_mapping=bool(get("kinetic_mapping", False)), commute_mapping=bool(get("commute_mapping", False)),
                **subclass_kwargs)
        F = get("n_features")
        if F is None or get("_outer_0_to_T_lagged") is None:
            return m                                  # an unfitted reference model: nothing to import
        F = int(F)
        c = np.ascontiguousarray(get("_outer_0_to_T_lagged"), dtype=np.float64)
        g = np.ascontiguousarray(np.asarray(get("_outer_0_to_TminusTau"), dtype=np.float64)
                                 + np.asarray(get("_outer_offset_to_T"), dtype=np.float64))
        s0 = np.ascontiguousarray(get("_sum_0_to_TminusTau"), dtype=np.float64)
        st = np.ascontiguousarray(get("_sum_tau_to_T"), dtype=np.float64)
        if c.shape != (F, F) or g.shape != (F, F) or s0.shape != (F,) or st.shape != (F,):
            raise ValueError("reference state does not describe a %d-feature model" % F)
        m._initialize(F)                              # creates the handle (zeroed)
        m.n_observations_ = int(get("n_observations_") or 0)
        m.n_sequences_ = int(get("n_sequences_") or 0)
        check(_lib.lib().msm_tica_import(m._handle, c.ctypes.data, g.ctypes.data, s0.ctypes.data, st.ctypes.data,
                                         m.n_observations_, m.n_sequences_))
        m._host_stale = True                          # the host mirrors are refreshed from the device on demand
        m._is_dirty = True
        m._mu_raw = None
        return m

    # --------------------------------------------------------------------- solve
    def _solve(self):
        """Top ``n_components`` generalized eigenpairs of (offset_correlation_, covariance_), cached until the
        accumulators change (tica.py:167-199).  Three routes: the finalisation of the moments, the shrinkage estimate, the
        Cholesky reduction, a subspace iteration for the top pairs and the back-substitution on the device
        (``msm_tica_solve_topk``: up to 16 components of 128 .. 1,024 features); the same with LAPACK's dsyevr on the
        reduced matrix where that does not apply or does not converge (``msm_tica_reduce`` / ``msm_tica_backsolve``);
        rocSOLVER from ``_moments.DEVICE_SOLVE_MIN_FEATURES`` (``msm_tica_solve_device``).
        ``MSMBUILDER_AMD_DEVICE_SOLVE=0`` is the all-host numpy / dsygvx path of the reference."""
        if not self._is_dirty:
            # n_components may have been raised since the last solve
            if len(self._eigenvalues_) >= self.n_components:
                return

        if not self.n_observations_:
            raise RuntimeError('The model must be fit() before use.')

        mode = _moments.solve_mode(self.n_features)
        if mode == "host":
            lhs = self.offset_correlation_
            rhs = self.covariance_
            if not _moments.is_symmetric(lhs):
                raise RuntimeError('offset correlation matrix is not symmetric')
            if not _moments.is_symmetric(rhs):
                raise RuntimeError('correlation matrix is not symmetric')
            vals, vecs = _moments.top_generalized_eigenpairs(lhs, rhs, self.n_components)
            self._mu_raw = None      # only the device solve leaves a mean behind; never keep one from an older state
        else:
            vals, vecs = self._solve_on_device(mode)

        self._eigenvalues_ = vals
        self._eigenvectors_ = vecs

        self._is_dirty = False

    def _solve_on_device(self, mode):
        self._ensure_handle()
        L = _lib.lib()
        F, k = int(self.n_features), int(self.n_components)
        if k > F:
            raise ValueError("Requested eigenvalue indices are not valid. Valid range is [0, %d] and start <= end, but "
                             "start=%d, end=%d is given" % (F - 1, F - k, F - 1))
        shrink = -1.0 if self.shrinkage is None else float(self.shrinkage)
        scale = getattr(self, "_input_scale", None)
        scale_p = None if scale is None else np.ascontiguousarray(scale, dtype=np.float64)
        mu = np.empty(F)
        info = np.zeros(12)

        def run(rc):
            if rc == _lib.MSM_ERR_NONFINITE:
                raise RuntimeError(_lib.last_error())      # the reference's np.allclose(lhs, lhs.T) checks, tica.py:183-186
            if rc == _lib.MSM_ERR_INVALID and "positive definite" in _lib.last_error():
                raise np.linalg.LinAlgError(_lib.last_error())
            check(rc)

        if mode == "device":
            vals = np.empty(k)
            vecs = np.empty((k, F))
            run(L.msm_tica_solve_device(self._handle, shrink, int(self.n_observations_),
                                        None if scale_p is None else scale_p.ctypes.data, k, vals.ctypes.data,
                                        vecs.ctypes.data, mu.ctypes.data, info.ctypes.data))
            V = vecs
        else:
            Cs = np.empty((F, F))
            done = False
            V = None
            if _moments.use_subspace_solve(F, k):
                # reduction + Chebyshev-filtered subspace iteration + L^-T on the device (csrc/subspace.hip); the pairs are
                # verified against the reduced matrix there -- status != 0 (no convergence on a flat spectrum, or a failed
                # check) hands Cs to the LAPACK route below
                vals = np.empty(k)
                V = np.empty((k, F))
                status = C.c_int(0)
                run(L.msm_tica_solve_topk(self._handle, shrink, int(self.n_observations_),
                                          None if scale_p is None else scale_p.ctypes.data, k, vals.ctypes.data, V.ctypes.data,
                                          Cs.ctypes.data, mu.ctypes.data, info.ctypes.data, C.byref(status)))
                self._solve_route = ("subspace" if status.value == 0 else "lapack", int(info[9]), status.value)
                if status.value == 0:
                    done = True
                else:
                    V = None
            else:
                run(L.msm_tica_reduce(self._handle, shrink, int(self.n_observations_),
                                      None if scale_p is None else scale_p.ctypes.data, Cs.ctypes.data, mu.ctypes.data,
                                      info.ctypes.data))
                self._solve_route = ("lapack", 0, 0)
            if not done:   # host dsyevr on the reduced matrix: the verified fallback
                vals, Y = _moments.top_standard_eigenpairs(Cs, k)   # Y: k x F, rows = eigenvectors of the reduced problem
            if V is None:
                V = np.empty((k, F))
                check(L.msm_tica_backsolve(self._handle, Y.ctypes.data, k, V.ctypes.data))
        self.shrinkage_ = float(info[0]) if self.shrinkage is None else self.shrinkage
        self._mu_raw = mu
        return vals, np.ascontiguousarray(V.T)

    @property
    def score_(self):
        """Training score: the sum of the first `n_components` eigenvalues"""
        self._solve()
        return self._eigenvalues_[:self.n_components].sum()

    @property
    def eigenvectors_(self):
        self._solve()
        return self._eigenvectors_[:, :self.n_components]

    @property
    def eigenvalues_(self):
        self._solve()
        return self._eigenvalues_[:self.n_components]

    @property
    def timescales_(self):
        self._solve()
        return -1. * self.lag_time / np.log(self._eigenvalues_[:self.n_components])

    @property
    def components_(self):
        return self.eigenvectors_[:, 0:self.n_components].T

    @property
    def _n_pairs(self):
        return _moments.pair_count(self.n_observations_, self.n_sequences_, self.lag_time)

    def set_input_scaling(self, shift=None, scale=None):
        """Treat every input row as ``(x - shift) / scale`` (per-column float64 arrays; ``None``
        removes the map).  The accumulators keep the RAW sums; since the centred moments
        transform as ``mu' = (mu - shift) / scale``, ``OC' = D OC D`` and ``S' = D S D`` with
        ``D = diag(1 / scale)``, the model is exactly the one a fit on the scaled data would
        give (up to rounding), ``transform`` keeps taking raw rows, and no scaled copy of the
        data set is ever written (the StandardScaler -> tICA pipeline of the reference's
        workflows)."""
        if shift is None and scale is None:
            self._input_shift = self._input_scale = None
        else:
            F = self.n_features if self.n_features is not None else len(np.atleast_1d(shift if shift is not None else scale))
            self._input_shift = np.zeros(F) if shift is None else np.array(np.broadcast_to(shift, (F,)), dtype=np.float64)
            self._input_scale = np.ones(F) if scale is None else np.array(np.broadcast_to(scale, (F,)), dtype=np.float64)
        self._is_dirty = True
        return self

    def _raw_means(self):
        if getattr(self, "_mu_raw", None) is not None:
            return self._mu_raw                 # left behind by the device-side solve of the CURRENT accumulators
                                                # (every accumulate / import / all-reduce clears it)
        if self._host_stale and self._handle is not None:
            # the two column sums alone: no need to download the F x F moments for a mean
            s0, st = np.empty(self.n_features), np.empty(self.n_features)
            check(_lib.lib().msm_tica_export_sums(self._handle, s0.ctypes.data, st.ctypes.data))
            return _moments.mean_vector(s0, st, self._n_pairs)
        self._pull()
        return _moments.mean_vector(self._sum_0_to_TminusTau, self._sum_tau_to_T, self._n_pairs)

    @property
    def means_(self):
        mu = self._raw_means()
        if getattr(self, "_input_scale", None) is not None:
            mu = (mu - self._input_shift) / self._input_scale
        return mu

    @property
    def offset_correlation_(self):
        self._pull()
        oc = _moments.offset_correlation(self._outer_0_to_T_lagged, self._raw_means(), self._n_pairs)
        if getattr(self, "_input_scale", None) is not None:
            oc = oc / np.outer(self._input_scale, self._input_scale)
        return oc

    @property
    def covariance_(self):
        """Shrunk covariance; reading it also sets ``shrinkage_`` (as in the reference)."""
        self._pull()
        S = _moments.sample_covariance(self._outer_gram_sum, self._raw_means(), self._n_pairs)
        if getattr(self, "_input_scale", None) is not None:
            S = S / np.outer(self._input_scale, self._input_scale)
        if self.shrinkage is None:
            self.shrinkage_ = _moments.rblw_shrinkage(S, n=self.n_observations_)
        else:
            self.shrinkage_ = self.shrinkage
        return _moments.shrink(S, self.shrinkage_)

    # ----------------------------------------------------------------------- fit
    def fit(self, sequences, y=None):
        """Forget everything seen before, then accumulate every trajectory of ``sequences`` (a list, or a lazily
        loading iterable, of (n_frames_i, n_features) arrays; numpy arrays are staged over PCIe, torch CUDA tensors
        are read in place).  Trajectories not longer than ``lag_time`` are skipped with a warning; if none is long
        enough a ValueError is raised.  ``y`` is ignored.  Use ``partial_fit`` to add data to an existing model.
        Returns ``self``."""
        self._initialized = False
        check_iter_of_sequences(sequences, max_iter=3)  # we might be lazy-loading
        # device-resident trajectories need no staging: up to 4096 of them share one launch
        group, group_bytes = [], 0
        # a list of views of ONE device tensor lying back to back (X.view(n, T, F).unbind(0)): one vectorised check instead
        # of a Python loop over the trajectories here and another in _fit_many (0.4 ms of a 1,000-trajectory fit)
        if isinstance(sequences, (list, tuple)) and 1 < len(sequences) <= 4096 and is_device_array(sequences[0]) \
                and sequences[0].dim() == 2 and sequences[0].shape[0] > 0:
            try:
                rows_adj = _lib.adjacent_rows(sequences)
            except Exception:
                rows_adj = None
            if rows_adj is not None:
                self._fit_many(sequences, rows_adj=rows_adj)
                sequences = ()
        # (a materialised list of host arrays goes down whole: the library stages it in 512 MiB groups through two buffers
        #  and copies group g + 1 while group g is accumulated -- splitting it here would put a synchronisation between the
        #  groups; a lazily loading iterable is consumed ~1 GiB at a time so that it is never held in memory at once)
        #  -- and so is a list whose members need a converted copy in _prepare (non-contiguous, float16, integers, lists of
        #  lists): those copies are held together for one _fit_many call, so only ~1 GiB of them goes into one call)
        lazy = not isinstance(sequences, (list, tuple))
        for X in sequences:
            group.append(X)
            if not getattr(X, "is_cuda", False):
                conforming = (isinstance(X, np.ndarray) and X.ndim == 2 and X.flags.c_contiguous
                              and X.dtype in (np.float32, np.float64))
                if lazy or not conforming:
                    group_bytes += int(np.prod(np.shape(X))) * int(getattr(getattr(X, "dtype", None), "itemsize", 8) or 8)
            if group_bytes >= _BATCH_BYTES or len(group) >= 4096:
                self._fit_many(group)
                group, group_bytes = [], 0
        if group:
            self._fit_many(group)

        if not self.n_sequences_:
            raise ValueError('All sequences were shorter than '
                             'the lag time, %d' % self.lag_time)

        return self

    def partial_fit(self, X):
        """Fit the model with X (online; the state is updated with the new data)."""
        self._fit(X)
        return self

    def _prepare(self, X, keep_bf16=False):
        """array2d + dtype rule: float32/float64 are consumed natively, anything else is
        up-cast to float64 exactly like tica.py:402.  The finite check of array2d (validation.py:68-74) is NOT run
        on the host for data that goes to the device: the column-sum kernel performs it on every frame it reads
        and the call raises the same ValueError (a numpy reduction over the trajectories was 2/3 of a host-array
        fit: 0.2 s per 4 GB against 0.1 s for upload + kernels)."""
        X = array2d(X, force_all_finite=False)
        if is_device_array(X):
            import torch
            if X.dtype == torch.bfloat16:
                # bf16-STORED trajectories (BASELINE configs[4]: half the bytes) feed the bf16 modes and the projection
                # kernels (keep_bf16 = "always") as they are; the other modes take them as float32 (an exact widening)
                # (the mode is the HANDLE's, fixed when it was created; the environment may have changed since)
                hk = getattr(self, "_handle_key", None)
                mode = hk[2] if (hk is not None and self._initialized) else _mode_from_env()
                if not (keep_bf16 == "always" or (keep_bf16 and mode in (_lib.TICA_BF16, _lib.TICA_BF16X2))):
                    X = X.to(torch.float32)
            elif X.dtype not in (torch.float32, torch.float64):
                X = X.to(torch.float64)
            return X.contiguous()
        if X.dtype not in (np.float32, np.float64):
            X = np.asarray(X, dtype=np.float64)
        return np.ascontiguousarray(X)

    def _fit(self, X):
        self._fit_many([X])

    def _fit_many(self, Xs, rows_adj=None):
        """One launch for a group of trajectories (tica.py:401-424 per trajectory).  ``rows_adj``: the row counts when the
        caller has already established that the group is back-to-back views of one device tensor."""
        prepared = []
        try:
            import torch
            fast_types = (torch.float32, torch.float64)
            Tensor = torch.Tensor
        except Exception:  # numpy-only hosts
            Tensor, fast_types = (), ()
        lag = self.lag_time
        # all-fast group (what a 10M-frame fit of device tensors is: 1,000 of them): 2-D contiguous CUDA tensors of ONE fast
        # dtype and the model's width, each long enough -- one pass collects pointers and lengths, nothing else is touched
        # per trajectory (the general path below costs ~1.1 us per trajectory, 1.1 ms of such a fit)
        if Tensor and len(Xs) and type(Xs[0]) is Tensor and Xs[0].dim() == 2:
            F = self.n_features if self._initialized else Xs[0].shape[1]
            dt0 = Xs[0].dtype
            if dt0 in fast_types and Xs[0].is_cuda:
                dev0 = Xs[0].device
                ptrs, nrows, ok = [], [], True
                # views of one tensor lying back to back (X.view(n, T, F).unbind(0), slices of a joined tensor): rows and
                # pointers from a few vectorised checks instead of six checks per trajectory (0.8 ms of a 1,000-trajectory fit)
                if rows_adj is not None and Xs[0].shape[1] != F:
                    rows_adj = None
                if rows_adj is None and len(Xs) > 1 and Xs[0].shape[0] > 0 and Xs[0].shape[1] == F:
                    try:
                        rows_adj = _lib.adjacent_rows(Xs)
                    except Exception:
                        rows_adj = None
                if rows_adj is not None and int(rows_adj.min()) > lag and int(rows_adj.min()) >= F:
                    step = F * Xs[0].element_size()
                    starts = Xs[0].data_ptr() + np.concatenate(([0], np.cumsum(rows_adj[:-1]))) * step
                    # the two tables go to the library as they are (a ctypes array built from a thousand Python integers
                    # costs 0.2 ms a piece)
                    ptrs = np.ascontiguousarray(starts, dtype=np.uint64)
                    nrows = np.ascontiguousarray(rows_adj, dtype=np.int64)
                for X in (Xs if not len(ptrs) else ()):
                    if type(X) is not Tensor:
                        ok = False
                        break
                    sh = X.shape
                    if (len(sh) != 2 or sh[1] != F or sh[0] <= lag or sh[0] < F or X.dtype is not dt0 or X.device != dev0
                            or X.stride() != (F, 1)):
                        ok = False
                        break
                    ptrs.append(X.data_ptr())
                    nrows.append(sh[0])
                if ok:
                    n = len(ptrs)
                    self._initialize(F)
                    self._ensure_handle()
                    _lib.ensure_device(dev0.index)
                    _lib.set_stream(torch.cuda.current_stream(dev0).cuda_stream)
                    skipped = C.c_int64(0)
                    if isinstance(ptrs, np.ndarray):
                        cptrs, crows = ptrs.ctypes.data_as(C.POINTER(C.c_void_p)), nrows.ctypes.data_as(C.POINTER(C.c_int64))
                    else:
                        cptrs, crows = (C.c_void_p * n)(*ptrs), (C.c_int64 * n)(*nrows)
                    check(_lib.lib().msm_tica_accumulate_batch(self._handle, cptrs, crows, n,
                                                               8 if dt0 is torch.float64 else 4, int(F), 1, 1, C.byref(skipped)))
                    self.n_observations_ += int(sum(nrows)) if not isinstance(nrows, np.ndarray) else int(nrows.sum())
                    self.n_sequences_ += n
                    self._host_stale = True
                    self._is_dirty = True
                    self._mu_raw = None
                    return
        for X in Xs:
            # fast path: a 2-D contiguous float32 / float64 CUDA tensor of the model's width that is long enough needs
            # none of the conversions below (a 10M-frame fit is 1,000 of them: the generic path costs 2 us each)
            if (type(X) is Tensor and X.is_cuda and X.dim() == 2 and X.dtype in fast_types and self._initialized
                    and X.shape[1] == self.n_features and X.shape[0] > lag and X.shape[0] >= X.shape[1]
                    and X.is_contiguous()):
                prepared.append(X)
                continue
            X = self._prepare(X, keep_bf16=True)
            if X.shape[1] > X.shape[0]:
                warnings.warn("The number of features (%d) is greater than the length of the data (%d). "
                              "The covariance matrix is not guaranteed to be positive definite."
                              % (X.shape[1], X.shape[0]))
            self._initialize(X.shape[1])
            if X.shape[1] != self.n_features:
                raise ValueError("shapes (%d,%d) and (%d,%d) not aligned" % (
                    self.n_features, self.n_features, X.shape[1], X.shape[0]))
            # We don't need to scream and shout here. Just ignore this data.
            if not len(X) > self.lag_time:
                if not is_device_array(X):
                    _assert_all_finite(X)   # never reaches the device: keep the reference's check (tica.py:402 -> array2d)
                warnings.warn("length of data (%d) is too short for the lag time (%d)"
                              % (len(X), self.lag_time))
                continue
            prepared.append(X)
        if not prepared:
            return
        self._ensure_handle()
        # one launch per (placement, dtype) class, preserving the reference's skip semantics
        classes = {}
        key_of = {}
        for X in prepared:
            dt = X.dtype
            key = key_of.get(dt)
            if key is None or key[0] != (type(X) is not np.ndarray):
                key = (is_device_array(X), 8 if str(dt).endswith("64") else 2 if str(dt).endswith("bfloat16") else 4)
                key_of[dt] = key
            classes.setdefault(key, []).append(X)
        L = _lib.lib()
        for (on_dev, nbytes), arrs in classes.items():
            n = len(arrs)
            if on_dev:
                # one Arr() for the device/stream binding, raw pointers for the rest (1000 trajectories
                # per call: per-item wrappers cost more than the launch)
                views = arrs
                import torch
                _lib.ensure_device(arrs[0].device.index)
                _lib.set_stream(torch.cuda.current_stream(arrs[0].device).cuda_stream)
                ptrs = (C.c_void_p * n)(*[a.data_ptr() for a in arrs])
            else:
                views = [Arr(a) for a in arrs]
                ptrs = (C.c_void_p * n)(*[v.ptr for v in views])
            nrows = [v.shape[0] for v in views]
            rows = (C.c_int64 * n)(*nrows)
            skipped = C.c_int64(0)
            check(L.msm_tica_accumulate_batch(self._handle, ptrs, rows, n, nbytes,
                                              int(self.n_features), int(on_dev), 1, C.byref(skipped)))
            self.n_observations_ += sum(nrows)
            self.n_sequences_ += n
        self._host_stale = True
        self._is_dirty = True
        self._mu_raw = None

    # ------------------------------------------------------------------ multi-GPU
    def partial_fit_segments(self, pieces):
        """Accumulate slices of trajectories: ``pieces`` is a list of ``(X_slice, length,
        row_offset, own_begin, own_end)`` where ``X_slice[0]`` is row ``row_offset`` of a
        trajectory of ``length`` rows and this model owns the LEFT frames ``[own_begin,
        own_end)`` of its lagged pairs (the slice must reach row ``min(own_end + lag_time,
        length) - 1``).  Summed over the models that share a trajectory (``allreduce``), the
        state equals ``partial_fit`` of the whole trajectory (tica.py:401-424)."""
        classes = {}
        for X, length, off, ob, oe in pieces:
            X = self._prepare(X)
            self._initialize(X.shape[1])
            if X.shape[1] != self.n_features:
                raise ValueError("shapes (%d,%d) and (%d,%d) not aligned" % (
                    self.n_features, self.n_features, X.shape[1], X.shape[0]))
            if not length > self.lag_time:
                continue
            key = (is_device_array(X), 8 if str(X.dtype).endswith("64") else 4)
            classes.setdefault(key, []).append((X, int(length), int(off), int(ob), int(oe)))
        if not classes:
            return self
        self._ensure_handle()
        L = _lib.lib()
        for (on_dev, nbytes), items in classes.items():
            n = len(items)
            views = [Arr(it[0]) for it in items]
            ptrs = (C.c_void_p * n)(*[v.ptr for v in views])
            rows = (C.c_int64 * n)(*[v.shape[0] for v in views])
            seg4 = (C.c_int64 * (4 * n))(*[x for it in items for x in it[1:]])
            skipped = C.c_int64(0)
            check(L.msm_tica_accumulate_segments(self._handle, ptrs, rows, seg4, n, nbytes,
                                                 int(self.n_features), int(on_dev), 1, C.byref(skipped)))
            for it in items:
                self.n_observations_ += max(0, it[4] - it[3])
                self.n_sequences_ += 1 if (it[3] == 0 and it[4] > it[3]) else 0
        self._host_stale = True
        self._is_dirty = True
        self._mu_raw = None
        return self

    def fit_sharded(self, sequences, group=None):
        """SPMD ``fit``: every rank of ``torch.distributed`` calls this with the SAME list of
        trajectories (arrays, or lazily loaded objects supporting ``len`` and row slicing);
        each rank reads only its balanced share of the frame axis -- long trajectories are cut
        between ranks with a ``lag_time``-row right halo (``parallel.split_frames``) -- and ONE
        all-reduce leaves the global model on every rank.  With a single process this is
        ``fit``."""
        from .. import parallel
        self._initialized = False
        check_iter_of_sequences(sequences, max_iter=3)
        lengths = [len(s) for s in sequences]
        for n in lengths:
            if not n > self.lag_time:
                warnings.warn("length of data (%d) is too short for the lag time (%d)" % (n, self.lag_time))
        if not any(n > self.lag_time for n in lengths):
            raise ValueError('All sequences were shorter than '
                             'the lag time, %d' % self.lag_time)
        pieces = []
        for i, ob, oe in parallel.split_frames(lengths, self.lag_time):
            end = min(oe + self.lag_time, lengths[i])
            pieces.append((sequences[i][ob:end], lengths[i], ob, ob, oe))
        if pieces:
            self.partial_fit_segments(pieces)
        else:  # more ranks than frames: contribute zeros
            self._initialize(int(np.shape(sequences[0])[1]))
            self._ensure_handle()
        if parallel.active():
            self.allreduce(group=group)
        return self

    def allreduce(self, group=None):
        """Sum the accumulators over all ranks of ``torch.distributed`` (RCCL over xGMI
        with the ``nccl`` backend; ``gloo`` on CPU-only test runs): one all-reduce of the
        packed [C | G | s0 | stau | n_obs | n_seq] buffer.  Every rank ends with the
        global model.  Call after the local ``fit``/``partial_fit`` calls."""
        from ..parallel import allreduce_tica
        allreduce_tica(self, group=group)
        return self

    # ------------------------------------------------------------------ transform
    def _projection(self):
        """(mean, k x F matrix) with the kinetic / commute column scaling folded in
        (tica.py:335-351)."""
        comps = np.array(self.components_, dtype=np.float64)
        if self.kinetic_mapping:
            comps = comps * self.eigenvalues_[:, None]
        if self.commute_mapping:
            with np.errstate(all="ignore"):
                regularized_timescales = 0.5 * self.timescales_ * \
                    np.tanh(np.pi * ((self.timescales_ - self.lag_time) / self.lag_time) + 1)
                scale = np.sqrt(regularized_timescales / 2)
            # the reference multiplies and then nan_to_num()s the result: a NaN scale
            # (negative eigenvalue / timescale below the lag) zeroes the column
            scale = np.where(np.isnan(scale), 0.0, scale)
            comps = comps * scale[:, None]
        if getattr(self, "_input_scale", None) is not None:
            # ((x - m)/s - (mu - m)/s) . V  ==  (x - mu) . (V / s): raw rows, raw mean, D folded into V
            comps = comps / self._input_scale[None, :]
        return np.ascontiguousarray(self._raw_means(), dtype=np.float64), np.ascontiguousarray(comps)

    def transform(self, sequences):
        """Project every trajectory onto the slow coordinates: one (n_frames_i, n_components) float64 array per input
        (tica.py:329-352), on the device when the input is.  The kinetic / commute scalings are folded into the
        projection matrix."""
        check_iter_of_sequences(sequences, max_iter=3)  # we might be lazy-loading
        # trajectories that lie back to back in one allocation are projected in ONE launch and the result cut per
        # trajectory (a launch per 10,000-frame trajectory keeps a sixth of the GPU busy)
        joined = _lib.adjacent_view(sequences) if isinstance(sequences, (list, tuple)) else None
        if joined is not None:
            return _lib.cut_rows(self.transform([joined])[0], [X.shape[0] for X in sequences])
        batched = self._transform_device_list(sequences) if isinstance(sequences, (list, tuple)) else None
        if batched is not None:
            return batched
        batched = self._transform_host_list(sequences) if isinstance(sequences, (list, tuple)) else None
        if batched is not None:
            return batched
        sequences_new = []
        mean, comps = None, None
        L = _lib.lib()
        for X in sequences:
            X = self._prepare(X, keep_bf16="always")
            if mean is None:
                mean, comps = self._projection()
            if X.shape[1] != comps.shape[1]:
                raise ValueError("shapes (%d,%d) and (%d,%d) not aligned" % (
                    X.shape[0], X.shape[1], comps.shape[1], comps.shape[0]))
            k = comps.shape[0]
            if is_device_array(X) and str(X.dtype).endswith("bfloat16"):
                # bfloat16-stored rows: widened inside the projection kernel (2 bytes per value read from HBM)
                import torch
                _lib.ensure_device(X.device.index)
                _lib.set_stream(torch.cuda.current_stream(X.device).cuda_stream)
                out = torch.empty((X.shape[0], k), dtype=torch.float64, device=X.device)
                if X.shape[0] > 0:
                    check(L.msm_tica_project(C.c_void_p(X.data_ptr()), 2, X.shape[0], X.shape[1], X.shape[1],
                                             mean.ctypes.data, comps.ctypes.data, k, C.c_void_p(out.data_ptr()), 1, 1))
                sequences_new.append(out)
                continue
            ax = Arr(X)
            out = _lib.empty_like_placement(ax, (ax.shape[0], k), np.float64)
            aout = Arr(out, np.float64)
            if ax.shape[0] > 0:
                check(L.msm_tica_project(ax.vp, ax.dtype.itemsize, ax.shape[0], ax.shape[1], ax.shape[1],
                                         mean.ctypes.data, comps.ctypes.data, k, aout.vp, ax.on_device, 1))
            sequences_new.append(out)
        return sequences_new

    def _transform_host_list(self, sequences):
        """``transform`` of several HOST trajectories (C-contiguous float32 or float64 numpy arrays of the model's width) in one
        library call: staged over PCIe in overlapped groups, one [total, k] result cut per trajectory
        (``msm_tica_project_host_list``); None when the list does not qualify -- the caller then goes trajectory by
        trajectory (copy, kernel, copy back and a synchronisation each: half the link's rate)."""
        if len(sequences) < 2:
            return None
        head = sequences[0]
        if not isinstance(head, np.ndarray) or head.ndim != 2 or head.dtype not in (np.float32, np.float64):
            return None
        F = head.shape[1]
        for X in sequences:
            if (not isinstance(X, np.ndarray) or X.ndim != 2 or X.shape[1] != F or X.dtype != head.dtype
                    or not X.flags.c_contiguous):
                return None
        mean, comps = self._projection()
        if F != comps.shape[1]:
            raise ValueError("shapes (%d,%d) and (%d,%d) not aligned" % (head.shape[0], F, comps.shape[1], comps.shape[0]))
        k = comps.shape[0]
        n = len(sequences)
        lens = [int(X.shape[0]) for X in sequences]
        Y = np.empty((sum(lens), k), dtype=np.float64)
        _lib.ensure_device()
        xp = (C.c_void_p * n)(*[X.ctypes.data if X.shape[0] else None for X in sequences])
        rows = (C.c_int64 * n)(*lens)
        check(_lib.lib().msm_tica_project_host_list(xp, rows, n, head.dtype.itemsize, F, mean.ctypes.data, comps.ctypes.data, k,
                                                    Y.ctypes.data, 1))
        return _lib.cut_rows(Y, lens)

    def _transform_device_list(self, sequences):
        """``transform`` of several separately allocated device trajectories in ONE launch per 16 components
        (``msm_tica_project_batch``: a table of 256-row tiles), the result one [total, k] tensor cut per trajectory; None
        when the list does not qualify (host arrays, mixed dtypes, rows that are not whole 16-byte vectors) -- the caller
        then projects trajectory by trajectory."""
        if len(sequences) < 2 or not all(is_device_array(X) for X in sequences):
            return None
        import torch
        head = sequences[0]
        if head.dim() != 2 or head.dtype not in (torch.float32, torch.float64, torch.bfloat16):
            return None
        F = head.shape[1]
        nbytes = head.element_size()
        if (F * nbytes) % 16 != 0:
            return None
        for X in sequences:
            if X.dim() != 2 or X.shape[1] != F or X.dtype != head.dtype or X.device != head.device:
                return None
        seqs = [X if X.is_contiguous() else X.contiguous() for X in sequences]
        if any(X.shape[0] and X.data_ptr() % 16 for X in seqs):
            return None
        mean, comps = self._projection()
        if F != comps.shape[1]:
            raise ValueError("shapes (%d,%d) and (%d,%d) not aligned" % (head.shape[0], F, comps.shape[1], comps.shape[0]))
        k = comps.shape[0]
        _lib.ensure_device(head.device.index)
        _lib.set_stream(torch.cuda.current_stream(head.device).cuda_stream)
        lens = [int(X.shape[0]) for X in seqs]
        Y = torch.empty((sum(lens), k), dtype=torch.float64, device=head.device)
        outs = _lib.cut_rows(Y, lens)
        n = len(seqs)
        xp = (C.c_void_p * n)(*[X.data_ptr() if X.shape[0] else None for X in seqs])
        op = (C.c_void_p * n)(*[o.data_ptr() if o.shape[0] else None for o in outs])
        rows = (C.c_int64 * n)(*lens)
        check(_lib.lib().msm_tica_project_batch(xp, op, rows, n, nbytes, F, mean.ctypes.data, comps.ctypes.data, k, 1))
        return outs

    def partial_transform(self, features):
        """Apply the dimensionality reduction on a single featurized trajectory."""
        sequences = [features]
        return self.transform(sequences)[0]

    def fit_transform(self, sequences, y=None):
        """Fit the model with X and apply the dimensionality reduction on X.  A list of host (numpy) trajectories that fits
        into a quarter of the free device memory crosses PCIe ONCE: it is uploaded back to back into one device buffer
        (``msm_upload_list``), fitted and projected there, and the projection comes back as one array cut per trajectory
        -- ``fit`` followed by ``transform`` on host arrays would upload every frame twice."""
        staged = self._stage_host_list(sequences, self.n_components)
        if staged is None:
            self.fit(sequences)
            return self.transform(sequences)
        Xd, lens = staged
        views = _lib.cut_rows(Xd, lens)
        self.fit(views)
        Yd = self.transform([Xd])[0]                     # [total, k] float64 on the device
        Y = np.empty(tuple(Yd.shape), dtype=np.float64)
        if Y.size:
            check(_lib.lib().msm_memcpy_d2h(Y.ctypes.data, C.c_void_p(Yd.data_ptr()), Y.nbytes))
        return _lib.cut_rows(Y, lens)

    @staticmethod
    def _stage_host_list(sequences, n_components=None):
        """(device tensor [total, F], row counts) holding a list of C-contiguous float32 / float64 numpy trajectories of one
        width back to back, or None when the list does not qualify (other types, mixed dtypes / widths, torch without a
        device, the staged rows PLUS the [total, k] float64 projection that fit_transform allocates beside them -- k = F when
        ``n_components`` is None: twice the bytes of float32 rows -- beyond a quarter of the free device memory)."""
        if not isinstance(sequences, (list, tuple)) or len(sequences) < 1:
            return None
        head = sequences[0]
        if not isinstance(head, np.ndarray) or head.ndim != 2 or head.dtype not in (np.float32, np.float64):
            return None
        F = head.shape[1]
        for X in sequences:
            if (not isinstance(X, np.ndarray) or X.ndim != 2 or X.shape[1] != F or X.dtype != head.dtype
                    or not X.flags.c_contiguous):
                return None
        try:
            import torch
            if not torch.cuda.is_available():
                return None
            _lib.ensure_device()
            dev = torch.device("cuda", int(_lib._initialized_device or 0))   # the library's device (LOCAL_RANK-aware)
            free_bytes = torch.cuda.mem_get_info(dev)[0]
        except Exception:
            return None
        lens = [int(X.shape[0]) for X in sequences]
        total = sum(lens)
        nbytes = total * F * head.dtype.itemsize
        k = F if n_components is None else min(int(n_components), F)
        if total == 0 or nbytes + total * k * 8 > free_bytes // 4:
            return None
        try:
            Xd = torch.empty((total, F), dtype=torch.float32 if head.dtype == np.float32 else torch.float64, device=dev)
        except RuntimeError:   # (torch's out-of-memory error: the caller falls back to fit + transform, staged in groups)
            return None
        n = len(sequences)
        src = (C.c_void_p * n)(*[X.ctypes.data if X.shape[0] else None for X in sequences])
        nb = (C.c_int64 * n)(*[int(X.nbytes) for X in sequences])
        _lib.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        check(_lib.lib().msm_upload_list(C.c_void_p(Xd.data_ptr()), src, nb, n))
        return Xd, lens

    # ---------------------------------------------------------------------- score
    def score(self, sequences, y=None):
        """Generalized matrix Rayleigh quotient of this model's slow directions on OTHER data (tica.py:426-467):
        fit a second model to ``sequences``, express its lagged and instantaneous moments in the basis V of this
        model's eigenvectors, and return trace((V^T OC' V) (V^T Sigma' V)^-1) -- the sum of this model's
        eigenvalues if the new data had exactly the training statistics, lower when the directions do not
        transfer.  NaN when the projected covariance is singular."""
        assert self._initialized
        V = self.eigenvectors_

        other = self.__class__(n_components=self.n_components, lag_time=self.lag_time, shrinkage=self.shrinkage)
        for X in sequences:
            other.partial_fit(X)
        if getattr(self, "_input_scale", None) is not None:
            other.set_input_scaling(self._input_shift, self._input_scale)

        lagged = V.T @ other.offset_correlation_ @ V
        instantaneous = V.T @ other.covariance_ @ V
        try:
            return np.trace(lagged @ np.linalg.inv(instantaneous))
        except np.linalg.LinAlgError:
            return np.nan

    def summarize(self):
        """Some summary information."""
        # force shrinkage to be calculated
        self.covariance_

        return """time-structure based Independent Components Analysis (tICA)
-----------------------------------------------------------
n_components        : {n_components}
shrinkage           : {shrinkage}
lag_time            : {lag_time}
kinetic_mapping     : {kinetic_mapping}

Top 5 timescales :
{timescales}

Top 5 eigenvalues :
{eigenvalues}
""".format(n_components=self.n_components, lag_time=self.lag_time,
           shrinkage=self.shrinkage_, kinetic_mapping=self.kinetic_mapping,
           timescales=self.timescales_[:5], eigenvalues=self.eigenvalues_[:5])
