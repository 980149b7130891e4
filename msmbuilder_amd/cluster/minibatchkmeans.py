"""Mini-batch k-means on MI355X: drop-in for ``msmbuilder.cluster.MiniBatchKMeans``.

In the reference this class is a three-line subclass of scikit-learn's estimator
(msmbuilder/cluster/__init__.py:67-69): every number comes from scikit-learn, which the
reference leaves unpinned.  This module restates scikit-learn 1.7's algorithm
(sklearn/cluster/_kmeans.py: ``MiniBatchKMeans.fit`` :2046-2196, ``_mini_batch_step``
:1557-1676, ``_mini_batch_convergence`` :1963-2027, ``_random_reassign`` :2029-2043) with
the same constructor arguments and the SAME host-side ``RandomState`` call sequence
(validation subsample, init subsample, k-means++ draws, one ``randint`` batch per step,
``choice`` for starved-centre reassignment), so that a given ``random_state`` walks the
same minibatches as scikit-learn.  The arithmetic that scales with the data runs on
the GPU (msmbuilder_amd/csrc/kmeans.hip): nearest-centre labelling as an MFMA
contraction, the per-centre streaming-mean update, and the final full-data labelling.

Element type (round 6): scikit-learn works in the type of X -- float32 rows give float32
centres, counts and distances, EVERYTHING ELSE (float64, integers, lists) is computed in
float64 -- and the reference pipeline hands this class the float64 output of
``tICA.transform`` (msmbuilder/decomposition/tica.py:329-352).  So does this module:
float32 rows run on ``v_mfma_f32_32x32x2_f32``, float64 rows on ``v_mfma_f64_16x16x4_f64``
(``kmeans_label_f64_kernel``), with float64 centres / counts / k-means++ potentials, and
``cluster_centers_`` comes back in the rows' type.  (Rounds 1-5 narrowed float64 input
to float32.)

Parity definition (DESIGN.md "MiniBatchKMeans"): float32 rows: centres rtol 1e-4, inertia
rtol 1e-4, labels equal except where the two best squared distances tie to fp32 rounding;
float64 rows: centres rtol 1e-9, inertia rtol 1e-9, labels equal off exact ties.

Multi-GPU: with ``torch.distributed`` initialised, each rank owns a shard of the frames;
every step all ranks draw the same global batch, label their share, and one RCCL
all-reduce of [K x F sums | K counts | inertia] makes the update identical everywhere
(see msmbuilder_amd/parallel.py).
"""
import ctypes as C
import warnings

import numpy as np
from sklearn.base import ClusterMixin, TransformerMixin
from sklearn.utils import check_random_state

from .. import _lib
from .._lib import Arr, check, empty_like_placement, is_device_array
from ..base import BaseEstimator
from .base import MultiSequenceClusterMixin

__all__ = ['MiniBatchKMeans']


def _work_dtype(X):
    """The type scikit-learn computes in for this input (``validate_data(dtype=[np.float64, np.float32])``): float32 stays
    float32, everything else becomes float64.  Device tensors: float64 stays, every other type (float32, half, bfloat16:
    types scikit-learn never sees) is computed in float32."""
    if is_device_array(X):
        import torch
        return np.dtype(np.float64) if X.dtype == torch.float64 else np.dtype(np.float32)
    dt = getattr(X, "dtype", None)
    return np.dtype(np.float32) if dt == np.float32 else np.dtype(np.float64)


def _sfx(dtype):
    return "f64" if np.dtype(dtype) == np.float64 else "f32"


def _rows_to_host(ax, idx):
    """X[idx] as a host array of X's type (device X: gather kernel + small D2H)."""
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    if not ax.on_device:
        return np.ascontiguousarray(ax.keep[idx])
    import torch
    sel = ax.keep[torch.as_tensor(idx, device=ax.keep.device)]
    return sel.detach().cpu().numpy()


def label_inertia(X, centers):
    """(labels int32, inertia) of every row of X against ``centers`` -- the GPU counterpart of
    scikit-learn's ``_labels_inertia``.  X: numpy or torch CUDA rows, float32 or float64: the arithmetic runs in the
    rows' type (``_work_dtype``), the centres are converted to it."""
    ax = X if isinstance(X, Arr) else Arr(X, _work_dtype(X))
    centers = np.ascontiguousarray(centers, dtype=ax.dtype)
    labels = empty_like_placement(ax, (ax.shape[0],), np.int32)
    if ax.shape[0] == 0:
        return labels, 0.0   # an empty trajectory: nothing to label (and an empty device tensor has no address)
    al = Arr(labels, np.int32)
    inertia = C.c_double(0.0)
    fn = getattr(_lib.lib(), "msm_kmeans_label_" + _sfx(ax.dtype))
    check(fn(ax.vp, ax.shape[0], ax.shape[1], centers.ctypes.data, centers.shape[0], al.vp, C.byref(inertia), ax.on_device))
    return labels, float(inertia.value)


def kmeans_plusplus(X, n_clusters, random_state):
    """Greedy k-means++ seeding of the rows ``X`` (numpy or torch CUDA, float32 or float64), scikit-learn's
    ``_kmeans_plusplus`` draw for draw (sklearn/cluster/_kmeans.py:163-259): first centre by ``choice``, then
    ``2 + log(k)`` candidates per round drawn by inverse-CDF sampling of the current squared distances, keeping the
    candidate with the lowest potential.  The rounds run on the device (``msm_kmeans_plusplus_f32``, csrc/kpp.hip); the
    draws -- one ``choice`` and ``(k - 1) x (2 + log k)`` uniforms -- come from ``random_state`` in scikit-learn's order
    (legacy ``RandomState.uniform`` draws element by element, so one call for all rounds is the same stream as one call
    per round).  Returns the centres as a host array of the rows' type."""
    ax = X if isinstance(X, Arr) else Arr(X, _work_dtype(X))
    n_samples, n_features = ax.shape
    n_local_trials = 2 + int(np.log(n_clusters))
    sample_weight = np.ones(n_samples, dtype=ax.dtype)
    center_id = int(random_state.choice(n_samples, p=sample_weight / sample_weight.sum()))
    u = np.ascontiguousarray(random_state.uniform(size=(max(n_clusters - 1, 0), n_local_trials)), dtype=np.float64)
    centers = np.empty((n_clusters, n_features), dtype=ax.dtype)
    ids = np.empty(n_clusters, dtype=np.int64)
    fn = getattr(_lib.lib(), "msm_kmeans_plusplus_" + _sfx(ax.dtype))
    check(fn(ax.vp, n_samples, n_features, n_clusters, center_id, u.ctypes.data, n_local_trials, centers.ctypes.data,
             ids.ctypes.data, ax.on_device))
    return centers


class _MiniBatchKMeans(ClusterMixin, TransformerMixin):
    """Single-array mini-batch k-means with scikit-learn 1.7's constructor."""

    def __init__(self, n_clusters=8, init='k-means++', max_iter=100, batch_size=1024, verbose=0,
                 compute_labels=True, random_state=None, tol=0.0, max_no_improvement=10,
                 init_size=None, n_init='auto', reassignment_ratio=0.01):
        self.n_clusters = n_clusters
        self.init = init
        self.max_iter = max_iter
        self.batch_size = batch_size
        self.verbose = verbose
        self.compute_labels = compute_labels
        self.random_state = random_state
        self.tol = tol
        self.max_no_improvement = max_no_improvement
        self.init_size = init_size
        self.n_init = n_init
        self.reassignment_ratio = reassignment_ratio

    # -- parameter resolution: _kmeans.py:868-907, 1922-1960 --
    def _check_params_vs_input(self, n_samples):
        if n_samples < self.n_clusters:
            raise ValueError("n_samples=%d should be >= n_clusters=%d." % (n_samples, self.n_clusters))
        init_is_array = hasattr(self.init, "__array__") or is_device_array(self.init)
        self._n_init = self.n_init
        if self._n_init == "auto":
            if isinstance(self.init, str) and self.init == "k-means++":
                self._n_init = 1
            elif isinstance(self.init, str) and self.init == "random":
                self._n_init = 3
            elif callable(self.init):
                self._n_init = 3
            else:
                self._n_init = 1
        if init_is_array and self._n_init != 1:
            warnings.warn("Explicit initial center position passed: performing only one init in "
                          "MiniBatchKMeans instead of n_init=%d." % self._n_init, RuntimeWarning)
            self._n_init = 1
        self._batch_size = min(self.batch_size, n_samples)
        self._init_size = self.init_size
        if self._init_size is None:
            self._init_size = 3 * self._batch_size
            if self._init_size < self.n_clusters:
                self._init_size = 3 * self.n_clusters
        elif self._init_size < self.n_clusters:
            warnings.warn("init_size=%d should be larger than n_clusters=%d. Setting it to "
                          "min(3*n_clusters, n_samples)" % (self._init_size, self.n_clusters),
                          RuntimeWarning)
            self._init_size = 3 * self.n_clusters
        self._init_size = min(self._init_size, n_samples)
        if self.reassignment_ratio < 0:
            raise ValueError("reassignment_ratio should be >= 0, got %s instead." % self.reassignment_ratio)

    def _init_centroids(self, ax, shard, random_state):
        """_kmeans.py:955-1045 (the init subsample is drawn even for an explicit array)."""
        n_samples = shard.n_total
        Xs = None
        from .. import parallel as _par
        # k-means++ runs on the device: rows that already live there are gathered there (no trip through the host)
        on_dev = ax.on_device and not _par.active() and isinstance(self.init, str) and self.init == "k-means++"

        def rows(idx):
            if on_dev:
                import torch
                return ax.keep[torch.as_tensor(np.ascontiguousarray(idx, dtype=np.int64), device=ax.keep.device)]
            return self._rows(ax, shard, idx)
        if self._init_size is not None and self._init_size < n_samples:
            init_indices = random_state.randint(0, n_samples, self._init_size)
            if isinstance(self.init, str) or callable(self.init):
                Xs = rows(init_indices)
        elif isinstance(self.init, str) or callable(self.init):
            Xs = ax.keep if on_dev else self._rows(ax, shard, np.arange(n_samples))
        if isinstance(self.init, str) and self.init == "k-means++":
            centers = kmeans_plusplus(Xs, self.n_clusters, random_state)   # on the device (csrc/kpp.hip)
        elif isinstance(self.init, str) and self.init == "random":
            w = np.ones(len(Xs), dtype=Xs.dtype)
            seeds = random_state.choice(len(Xs), size=self.n_clusters, replace=False, p=w / w.sum())
            centers = Xs[seeds]
        elif callable(self.init):
            centers = np.asarray(self.init(Xs, self.n_clusters, random_state=random_state))
        else:
            init = self.init
            if is_device_array(init):
                init = init.detach().cpu().numpy()
            centers = np.array(init, dtype=ax.dtype, copy=True, order="C")
            if centers.shape != (self.n_clusters, ax.shape[1]):
                raise ValueError("The shape of the initial centers %s does not match the number of "
                                 "clusters %d / features %d." % (centers.shape, self.n_clusters, ax.shape[1]))
        return np.ascontiguousarray(centers, dtype=ax.dtype)

    def _random_reassign(self):
        """_kmeans.py:2029-2043"""
        self._n_since_last_reassign += self._batch_size
        if (self._counts == 0).any() or self._n_since_last_reassign >= (10 * self.n_clusters):
            self._n_since_last_reassign = 0
            return True
        return False

    def _mini_batch_convergence(self, step, n_steps, n_samples, centers_squared_diff, batch_inertia):
        """_kmeans.py:1963-2027"""
        batch_inertia /= self._batch_size
        step = step + 1
        if step == 1:
            if self.verbose:
                print("Minibatch step %d/%d: mean batch inertia: %s" % (step, n_steps, batch_inertia))
            return False
        if self._ewa_inertia is None:
            self._ewa_inertia = batch_inertia
        else:
            alpha = self._batch_size * 2.0 / (n_samples + 1)
            alpha = min(alpha, 1)
            self._ewa_inertia = self._ewa_inertia * (1 - alpha) + batch_inertia * alpha
        if self.verbose:
            print("Minibatch step %d/%d: mean batch inertia: %s, ewa inertia: %s"
                  % (step, n_steps, batch_inertia, self._ewa_inertia))
        if self._tol > 0.0 and centers_squared_diff <= self._tol:
            if self.verbose:
                print("Converged (small centers change) at step %d/%d" % (step, n_steps))
            return True
        if self._ewa_inertia_min is None or self._ewa_inertia < self._ewa_inertia_min:
            self._no_improvement = 0
            self._ewa_inertia_min = self._ewa_inertia
        else:
            self._no_improvement += 1
        if self.max_no_improvement is not None and self._no_improvement >= self.max_no_improvement:
            if self.verbose:
                print("Converged (lack of improvement in inertia) at step %d/%d" % (step, n_steps))
            return True
        return False

    def _step(self, ax, shard, batch_idx, random_state, random_reassign):
        """One ``_mini_batch_step`` (_kmeans.py:1557-1676) on the device-resident state
        (``self._mbk``): label + streaming-mean update on the GPU, starved-centre reassignment
        decided on the host with scikit-learn's RNG calls.  ``batch_idx`` are GLOBAL row numbers
        (identical on every rank); each rank processes the rows it owns and, when sharded, one
        all-reduce of [sums | counts | inertia] makes the update identical everywhere.  Returns the
        batch inertia (computed before the update, as scikit-learn does)."""
        from .. import parallel
        K, F = self.n_clusters, ax.shape[1]
        B = len(batch_idx)
        inertia = C.c_double(0.0)
        L = _lib.lib()
        if parallel.active():
            # the rows of the batch this rank owns -> [K*F sums | K counts | inertia] in the handle's DEVICE buffer; one
            # all-reduce over the library communicator (msm_mbk_allreduce: RCCL on the library stream, in place) and the
            # identical update on every rank.  Only the batch's row numbers go in and [inertia | counts] come out.
            parallel.library_comm()
            _, sub = shard.local(batch_idx)
            sub = np.ascontiguousarray(sub, dtype=np.int64)
            if len(sub):
                check(L.msm_mbk_step(self._mbk, ax.vp, ax.shape[0], sub.ctypes.data, len(sub), C.byref(inertia),
                                     None, 0, ax.on_device))
            else:
                check(L.msm_mbk_zero_packed(self._mbk))
            check(L.msm_mbk_allreduce(self._mbk, C.byref(inertia), self._counts.ctypes.data))
            inertia_v = float(inertia.value)
        else:
            check(L.msm_mbk_step(self._mbk, ax.vp, ax.shape[0], batch_idx.ctypes.data, B, C.byref(inertia),
                                 self._counts.ctypes.data, 1, ax.on_device))
            inertia_v = float(inertia.value)

        if random_reassign:
            self._reassign(ax, shard, batch_idx, random_state)
        return inertia_v

    def _reassign(self, ax, shard, batch_idx, random_state):
        """The starved-centre part of ``_mini_batch_step`` (_kmeans.py:1620-1662), decided on the host from the
        counts the step returned, with scikit-learn's RNG calls."""
        L = _lib.lib()
        B = len(batch_idx)
        if self.reassignment_ratio > 0:
            weight_sums = self._counts
            to_reassign = weight_sums < self.reassignment_ratio * weight_sums.max()
            # pick at most .5 * batch_size samples as new centers
            if to_reassign.sum() > 0.5 * B:
                keep = np.argsort(weight_sums)[int(0.5 * B):]
                to_reassign[keep] = False
            n_reassigns = int(to_reassign.sum())
            if n_reassigns:
                self._drop_prefetch(random_state)   # the next run's indices were drawn ahead: this draw comes BEFORE them
                new_centers = random_state.choice(B, replace=False, size=n_reassigns)
                if self.verbose:
                    print("[MiniBatchKMeans] Reassigning %d cluster centers." % n_reassigns)
                # reset counts of reassigned centers, but don't reset them too small
                new_count = float(np.min(weight_sums[~to_reassign]))
                rows = np.ascontiguousarray(self._rows(ax, shard, batch_idx[new_centers]), dtype=ax.dtype)
                which = np.ascontiguousarray(np.nonzero(to_reassign)[0], dtype=np.int64)
                ridx = np.arange(n_reassigns, dtype=np.int64)
                check(L.msm_mbk_reassign(self._mbk, rows.ctypes.data, n_reassigns, ridx.ctypes.data,
                                         which.ctypes.data, n_reassigns, new_count, 0))
                weight_sums[to_reassign] = new_count
            else:
                mn = np.min(weight_sums[~to_reassign]) if (~to_reassign).any() else 0.0
                if to_reassign.any():
                    weight_sums[to_reassign] = mn
                    check(L.msm_mbk_set_counts(self._mbk, weight_sums.ctypes.data))

    def _drop_prefetch(self, random_state):
        """Indices drawn ahead for a run that will not happen: put the generator back in front of that draw."""
        pre = self.__dict__.pop("_prefetched", None)
        if pre is not None:
            random_state.set_state(pre[0])

    def _run(self, ax, shard, first_step, n_steps, n_samples, random_state):
        """Steps ``first_step ...`` up to and including the next one that looks at the counts for a random
        reassignment, queued on the device in one go (``msm_mbk_run``): the batches are drawn up front, the
        convergence bookkeeping of ``_mini_batch_convergence`` runs on the device after every step, and the host
        synchronises once per run instead of twice per step.  Returns (steps executed, converged).  Only the last step
        of a run may reassign, so the ``RandomState`` call sequence is scikit-learn's; if the criterion fires early the
        generator is rewound to where scikit-learn would have left it."""
        L = _lib.lib()
        B = self._batch_size

        def plan_from(step0, since):
            plan = []
            while step0 + len(plan) < n_steps and len(plan) < 256:
                since += B
                hit = since >= 10 * self.n_clusters
                plan.append(hit)
                if hit:
                    break
            return plan
        plan = plan_from(first_step, self._n_since_last_reassign)
        S = len(plan)
        # the S batches in ONE call: legacy RandomState.randint draws element by element from the bit stream, so this is
        # the same stream (and leaves the same state) as scikit-learn's S calls of size B -- checked by the n_steps_ /
        # generator-state comparisons against scikit-learn in the tests; S calls + S state snapshots cost 0.47 ms per run
        # on the host, more than the run's 0.23 ms on the GPU
        pre = self.__dict__.pop("_prefetched", None)
        if pre is not None and pre[1].shape == (S, B):
            state0, idx = pre          # drawn while the previous run executed; the generator already stands behind it
        else:
            if pre is not None:
                random_state.set_state(pre[0])
            state0 = random_state.get_state()
            idx = np.ascontiguousarray(random_state.randint(0, n_samples, (S, B)), dtype=np.int64)
        alpha = min(B * 2.0 / (n_samples + 1), 1)
        st = np.array([self._ewa_inertia or 0.0, self._ewa_inertia_min or 0.0, float(self._no_improvement),
                       0.0 if self._ewa_inertia is None else 1.0, 0.0 if self._ewa_inertia_min is None else 1.0, 0.0])
        done, conv = C.c_int64(0), C.c_int(0)
        inertias = np.empty(S)
        mni = -1 if self.max_no_improvement is None else int(self.max_no_improvement)
        from .. import parallel
        if parallel.active():
            # row-sharded: the same global batches on every rank; this rank's rows of each batch as local row numbers, one
            # library all-reduce per step inside msm_mbk_run_sharded (no Python, no host synchronisation inside the run)
            parallel.library_comm()
            flat = idx.ravel()
            mine = (flat >= shard.offset) & (flat < shard.offset + shard.n_local)
            local = np.ascontiguousarray(flat[mine] - shard.offset, dtype=np.int64)
            offs = np.zeros(S + 1, dtype=np.int64)
            np.cumsum(mine.reshape(S, B).sum(axis=1), out=offs[1:])
            check(L.msm_mbk_run_sharded(self._mbk, ax.vp, ax.shape[0], local.ctypes.data, offs.ctypes.data, S, B, first_step,
                                        alpha, mni, st.ctypes.data, C.byref(done), C.byref(conv), inertias.ctypes.data,
                                        self._counts.ctypes.data))
        else:
            check(L.msm_mbk_run_begin(self._mbk, ax.vp, ax.shape[0], idx.ctypes.data, S, B, first_step, alpha, mni, st.ctypes.data))
            # while the device runs: the NEXT run's batch indices, assuming this run completes (what it does but once per
            # fit).  The generator is left behind that draw; whoever needs it in between (the rewind below, a random
            # reassignment in `_reassign`) first puts it back to `_prefetched[0]`.
            nxt = plan_from(first_step + S, 0 if plan[-1] else self._n_since_last_reassign + S * B)
            if nxt:
                st_n = random_state.get_state()
                self._prefetched = (st_n, np.ascontiguousarray(random_state.randint(0, n_samples, (len(nxt), B)), dtype=np.int64))
            check(L.msm_mbk_run_end(self._mbk, st.ctypes.data, C.byref(done), C.byref(conv), inertias.ctypes.data,
                                    self._counts.ctypes.data))
        done = int(done.value)
        self._ewa_inertia = float(st[0]) if st[3] else None
        self._ewa_inertia_min = float(st[1]) if st[4] else None
        self._no_improvement = int(st[2])
        self._n_since_last_reassign += done * B
        if done == S and plan[-1]:
            self._n_since_last_reassign = 0
            self._reassign(ax, shard, idx[-1], random_state)
        elif done < S:  # the criterion fired early: leave the generator where scikit-learn would (rare: once per fit)
            self.__dict__.pop("_prefetched", None)
            random_state.set_state(state0)
            random_state.randint(0, n_samples, (done, B))
        return done, bool(conv.value)

    def _mbk_open(self, centers, counts):
        """Device state of the element type of ``centers`` (float32 / float64: the rows' type); ``counts`` must match."""
        assert centers.dtype == counts.dtype and centers.dtype in (np.float32, np.float64)
        h = C.c_void_p()
        create = _lib.lib().msm_mbk_create_f64 if centers.dtype == np.float64 else _lib.lib().msm_mbk_create
        check(create(C.byref(h), centers.shape[0], centers.shape[1]))
        self._mbk = h
        check(_lib.lib().msm_mbk_set(h, centers.ctypes.data, counts.ctypes.data))

    def _mbk_close(self):
        """Fetch the centres back and free the device state."""
        h = self.__dict__.pop("_mbk", None)
        if h is None:
            return None
        K, F = self.n_clusters, self.n_features_in_
        centers = np.empty((K, F), dtype=self._counts.dtype)
        try:
            check(_lib.lib().msm_mbk_get(h, centers.ctypes.data, self._counts.ctypes.data))
        finally:
            _lib.lib().msm_mbk_destroy(h)
        return centers

    @staticmethod
    def _rows(ax, shard, global_idx):
        """Rows of the (possibly row-sharded) data as a host array of the data's type."""
        return shard.gather_rows(lambda loc: _rows_to_host(ax, loc), global_idx, ax.shape[1], dtype=ax.dtype)

    def fit(self, X, y=None):
        ax = Arr(X, _work_dtype(X))   # float32 stays float32, everything else is computed in float64 (scikit-learn's rule)
        if len(ax.shape) != 2:
            raise ValueError("Expected 2D array")
        # indices drawn ahead by an earlier fit that ended in an exception belong to ITS data and ITS generator: never reuse them
        self.__dict__.pop("_prefetched", None)
        from ..parallel import RowShard
        shard = RowShard(ax.shape[0])  # single process: the whole array
        n_samples, n_features = shard.n_total, ax.shape[1]
        self._check_params_vs_input(n_samples)
        from .. import parallel as _par
        if _par.active() and not isinstance(self.random_state, (int, np.integer)):
            # every rank must draw the SAME validation / init / batch indices: with random_state=None (or a shared
            # RandomState object that differs per process) rank 0's draw seeds them all
            seed0 = check_random_state(self.random_state).randint(0, 2 ** 31 - 1) if _par.rank() == 0 else 0
            seed0 = int(_par.allreduce_array(np.array([float(seed0)]))[0])
            random_state = np.random.RandomState(seed0)
        else:
            random_state = check_random_state(self.random_state)
        self.n_features_in_ = n_features

        if self.tol > 0:
            if _par.active():
                # the variance of ALL rows (so every rank makes the same stop decision): all-reduced moments
                Xl = ax.keep.double() if ax.on_device else np.asarray(ax.keep, dtype=np.float64)
                s1 = np.asarray((Xl.sum(0).cpu().numpy() if ax.on_device else Xl.sum(0)), dtype=np.float64)
                s2 = np.asarray(((Xl * Xl).sum(0).cpu().numpy() if ax.on_device else (Xl * Xl).sum(0)), dtype=np.float64)
                red = _par.allreduce_array(np.concatenate([s1, s2]))
                mean = red[:n_features] / n_samples
                var = red[n_features:] / n_samples - mean * mean
                self._tol = float(np.mean(var)) * self.tol
            elif ax.on_device:
                self._tol = float(ax.keep.var(dim=0, unbiased=False).mean().item()) * self.tol
            else:
                self._tol = float(np.mean(np.var(ax.keep, axis=0))) * self.tol
        else:
            self._tol = 0.0

        # Validation set for the init
        validation_indices = random_state.randint(0, n_samples, self._init_size)
        X_valid = self._rows(ax, shard, validation_indices)

        best_inertia = None
        for init_idx in range(self._n_init):
            cluster_centers = self._init_centroids(ax, shard, random_state)
            _, inertia = label_inertia(X_valid, cluster_centers)
            if best_inertia is None or inertia < best_inertia:
                init_centers = cluster_centers
                best_inertia = inertia

        centers = np.ascontiguousarray(init_centers, dtype=ax.dtype)
        self._counts = np.zeros(self.n_clusters, dtype=ax.dtype)
        self._ewa_inertia = None
        self._ewa_inertia_min = None
        self._no_improvement = 0
        self._n_since_last_reassign = 0

        n_steps = (self.max_iter * n_samples) // self._batch_size
        i = -1
        self._mbk_open(centers, self._counts)
        import os
        from .. import parallel
        runs = (ax.on_device and not self.verbose and self._tol == 0.0
                and os.environ.get("MSMBUILDER_AMD_MBK_RUNS", "1") != "0")
        try:
            while i + 1 < n_steps:
                if runs and not (self._counts == 0).any():
                    done, converged = self._run(ax, shard, i + 1, n_steps, n_samples, random_state)
                    i += done
                else:  # one step at a time: starved centres (the first steps), tol > 0, verbose, host data
                    i += 1
                    self._drop_prefetch(random_state)
                    minibatch_indices = random_state.randint(0, n_samples, self._batch_size)
                    minibatch_indices = np.ascontiguousarray(minibatch_indices, dtype=np.int64)
                    if self._tol > 0.0:  # the tol criterion needs the centres on the host every step
                        prev = np.empty_like(centers)
                        check(_lib.lib().msm_mbk_get(self._mbk, prev.ctypes.data, None))
                    batch_inertia = self._step(ax, shard, minibatch_indices, random_state, self._random_reassign())
                    centers_squared_diff = 0
                    if self._tol > 0.0:
                        cur = np.empty_like(centers)
                        check(_lib.lib().msm_mbk_get(self._mbk, cur.ctypes.data, None))
                        centers_squared_diff = np.sum((cur - prev) ** 2)
                    converged = self._mini_batch_convergence(i, n_steps, n_samples, centers_squared_diff, batch_inertia)
                if converged:
                    break
        finally:
            # leave the generator where scikit-learn does -- also when a step raised (the tuple used to survive an
            # exception, with the caller's RandomState advanced past the prefetch)
            self._drop_prefetch(random_state)
            centers = self._mbk_close()

        self.cluster_centers_ = centers
        self.n_steps_ = i + 1
        self.n_iter_ = int(np.ceil(((i + 1) * self._batch_size) / n_samples))

        if self.compute_labels:
            # embarrassingly parallel: every rank labels its own rows; inertia is summed
            self.labels_, inertia = label_inertia(ax, self.cluster_centers_)
            from .. import parallel
            self.inertia_ = float(parallel.allreduce_array(np.array([inertia]))[0]) if parallel.active() else inertia
        else:
            self.inertia_ = self._ewa_inertia * n_samples
        return self

    def partial_fit(self, X, y=None):
        """One mini-batch update on X itself (sklearn ``MiniBatchKMeans.partial_fit``).  The first call fixes the element
        type (that of its X, by ``_work_dtype``); later batches are converted to it, like scikit-learn converts to the
        type of ``cluster_centers_``."""
        has_centers = hasattr(self, "cluster_centers_")
        ax = Arr(X, self.cluster_centers_.dtype if has_centers else _work_dtype(X))
        n_samples = ax.shape[0]
        if not has_centers:
            self._check_params_vs_input(n_samples)
            self._random_state = check_random_state(self.random_state)
            self._tol = 0.0
            self._batch_size = n_samples
            from ..parallel import RowShard
            self.cluster_centers_ = self._init_centroids(ax, RowShard(n_samples), self._random_state)
            self._counts = np.zeros(self.n_clusters, dtype=ax.dtype)
            self._n_since_last_reassign = 0
            self.n_steps_ = 0
        self._batch_size = n_samples
        from ..parallel import RowShard
        self.n_features_in_ = ax.shape[1]
        self._mbk_open(np.ascontiguousarray(self.cluster_centers_, dtype=ax.dtype), self._counts)
        try:
            self._step(ax, RowShard(n_samples), np.arange(n_samples, dtype=np.int64), self._random_state,
                       self._random_reassign())
        finally:
            self.cluster_centers_ = self._mbk_close()
        if self.compute_labels:
            self.labels_, self.inertia_ = label_inertia(ax, self.cluster_centers_)
        self.n_steps_ += 1
        return self

    def _like_centers(self, X):
        """X in the element type of the fitted centres (scikit-learn's ``_check_test_data`` converts to it)."""
        return Arr(X, self.cluster_centers_.dtype)

    def predict(self, X):
        """Index of the closest centre (squared euclidean, GEMM form, in the centres' element type) for each row of X."""
        labels, _ = label_inertia(self._like_centers(X), self.cluster_centers_)
        return labels

    def fit_predict(self, X, y=None):
        return self.fit(X).labels_

    def score(self, X, y=None):
        """Opposite of the k-means objective on X."""
        _, inertia = label_inertia(self._like_centers(X), self.cluster_centers_)
        return -inertia


class MiniBatchKMeans(MultiSequenceClusterMixin, _MiniBatchKMeans, BaseEstimator):
    __doc__ = """Mini-Batch K-Means clustering of a list of sequences (see module docstring).

    Parameters are scikit-learn's ``MiniBatchKMeans`` parameters; ``labels_`` is a list of
    int32 arrays, one per input sequence (msmbuilder/cluster/base.py:50-51).
    """

    def fit_predict(self, sequences, y=None):
        self.fit(sequences)
        return self.labels_

    def summarize(self):
        return """MiniBatchKMeans clustering
--------------------------
n_clusters : {n_clusters}
batch_size : {batch_size}
n_steps    : {n_steps}

Inertia    : {inertia}
""".format(n_clusters=self.n_clusters, batch_size=self.batch_size,
           n_steps=getattr(self, "n_steps_", None), inertia=getattr(self, "inertia_", None))
