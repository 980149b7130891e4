"""K-Centers (Gonzalez farthest-point) clustering on MI355X: drop-in for
``msmbuilder.cluster.KCenters`` (reference: msmbuilder/cluster/kcenters.py:24-182).

The reference runs ``n_clusters`` python iterations of ``libdistance.dist`` + numpy
mask/argmax passes on one CPU thread (kcenters.py:91-97).  Here the whole loop is
``msm_kcenters_fit_*`` in libmsmhip: one fused launch per centre (distance to the newest
centre, strict running-min update of ``distances_``/``labels_``, per-block argmax) with
no host round trip in between; ``predict`` is the exact ``assign_nearest`` kernel.
Results are bit-identical to the reference: same centre ids, labels and float64
distances, for float32 and float64 input and every vector metric.
"""
import ctypes as C

import numpy as np
from sklearn.base import ClusterMixin, TransformerMixin
from sklearn.utils import check_random_state

from .. import _lib, libdistance
from .._lib import Arr, check, empty_like_placement, is_device_array
from ..base import BaseEstimator
from .base import MultiSequenceClusterMixin

__all__ = ['KCenters']


def _as_float(X):
    """kcenters.py:80-82: anything that is not float32/float64 becomes float64."""
    if isinstance(X, np.ndarray):
        if not (X.dtype == 'float32' or X.dtype == 'float64'):
            X = X.astype('float64')
    elif is_device_array(X):
        import torch
        if X.dtype not in (torch.float32, torch.float64):
            X = X.to(torch.float64)
    return X


class _KCenters(ClusterMixin, TransformerMixin):
    """Farthest-point clustering of ONE array (the sequence-list estimator is :class:`KCenters`).

    Starting from a randomly drawn row, every further centre is the row that is currently farthest from
    all centres chosen so far; each row is labelled with its nearest centre.

    Parameters
    ----------
    n_clusters : int (default 8)
        How many centres to pick.
    metric : str (default "euclidean")
        One of libdistance's vector metrics: euclidean, sqeuclidean, cityblock, chebyshev, canberra,
        braycurtis, hamming, jaccard.  (The reference's "rmsd" needs mdtraj trajectories and is out of scope.)
    random_state : int, numpy RandomState or None
        Source of the first centre: ``check_random_state(random_state).randint(0, n_samples)``, the same
        draw as the reference's, so equal seeds give equal clusterings.

    Attributes
    ----------
    cluster_ids_ : list of n_clusters row indices, in the order the centres were chosen
    cluster_centers_ : (n_clusters, n_features) the rows themselves
    labels_ : (n_samples,) int64, index into cluster_ids_ of the nearest centre
    distances_ : (n_samples,) float64, distance to that centre
    inertia_ : float, sum of distances_
    """

    _force_sharded = False   # measurement hook, see fit()

    def __init__(self, n_clusters=8, metric='euclidean', random_state=None):
        self.n_clusters = n_clusters
        self.metric = metric
        self.random_state = random_state

    def fit(self, X, y=None):
        X = _as_float(X)
        metric = self.metric.decode() if isinstance(self.metric, bytes) else self.metric
        if metric not in libdistance.VECTOR_METRICS:
            raise ValueError('metric must be one of %s' %
                             ', '.join("'%s'" % s for s in libdistance.VECTOR_METRICS))
        from .. import parallel
        # _force_sharded (a class attribute bench.py's strong-scaling model sets): take the row-sharded library loop in a
        # single process too (a world of one; its all-gathers degenerate to copies)
        if parallel.active() or (self._force_sharded and is_device_array(X)):
            return self._fit_sharded(X, metric)
        n_samples = len(X)
        seed = check_random_state(self.random_state).randint(0, n_samples)
        ax = Arr(X)
        if len(ax.shape) != 2:
            raise ValueError("X must be 2-dimensional")
        kind = "f64" if ax.dtype == np.float64 else "f32"
        K = int(self.n_clusters)
        ids = np.zeros(K, dtype=np.int64)
        labels = empty_like_placement(ax, (n_samples,), np.int64)
        distances = empty_like_placement(ax, (n_samples,), np.float64)
        al, ad = Arr(labels, np.int64), Arr(distances, np.float64)
        inertia = C.c_double(0.0)
        # cluster_centers_ (kcenters.py:98: the chosen rows) comes back with the ids as a HOST array, like the reference's
        # attribute: gathering it afterwards with torch indexing was three more round trips, and `predict` needs it on
        # the host anyway
        centers = np.empty((K, ax.shape[1]), dtype=ax.dtype)
        fn = getattr(_lib.lib(), "msm_kcenters_fit2_" + kind)
        check(fn(ax.vp, n_samples, ax.shape[1], K, metric.encode(), int(seed), ids.ctypes.data,
                 al.vp, ad.vp, C.byref(inertia), ax.on_device, centers.ctypes.data))
        self.labels_ = labels
        self.distances_ = distances
        self.cluster_ids_ = ids.tolist()
        self.cluster_centers_ = centers
        # np.sum(distances_) as in kcenters.py:101 on the host; on the device the kernel's
        # fp64 tree sum of the same values
        self.inertia_ = np.sum(distances) if not ax.on_device else float(inertia.value)
        return self

    def _fit_sharded(self, X, metric):
        """Row-sharded k-centers (one process per GPU): X is THIS rank's block of rows, ranks own consecutive blocks
        of the global array.  The whole fit is ONE library call, ``msm_kcenters_fit_sharded_*``: per centre a pass
        kernel, the shard's candidate record ``[max distance | global row | that row's coordinates]``, one
        all-gather over the library communicator (RCCL over xGMI on the library stream) and a select kernel that
        picks the same winner on every rank -- nothing returns to the host or to Python inside the centre loop.
        Results equal the single-process fit of the concatenated data bit for bit (ties go to the lowest GLOBAL row,
        numpy's argmax).  labels_/distances_ stay sharded."""
        from .. import parallel
        import torch
        if isinstance(X, np.ndarray):
            X = torch.from_numpy(np.ascontiguousarray(X)).cuda()
        ax = Arr(X)
        n_local, m = ax.shape
        shard = parallel.RowShard(n_local)
        kind = "f64" if ax.dtype == np.float64 else "f32"
        K = int(self.n_clusters)
        rank = parallel.rank()
        # rank 0's draw decides (random_state=None would differ per process)
        seed = check_random_state(self.random_state).randint(0, shard.n_total)
        seed = int(parallel.allreduce_array(np.array([float(seed) if rank == 0 else 0.0]))[0])
        labels = empty_like_placement(ax, (n_local,), np.int64)
        distances = empty_like_placement(ax, (n_local,), np.float64)
        al, ad = Arr(labels, np.int64), Arr(distances, np.float64)
        parallel.library_comm()
        ids = np.zeros(K, dtype=np.int64)
        centers = np.zeros((K, m), dtype=ax.dtype)
        inertia = C.c_double(0.0)
        fn = getattr(_lib.lib(), "msm_kcenters_fit_sharded_" + kind)
        check(fn(ax.vp, n_local, m, K, metric.encode(), seed, shard.offset, al.vp, ad.vp, ids.ctypes.data,
                 centers.ctypes.data, C.byref(inertia)))
        self.labels_ = labels
        self.distances_ = distances
        self.cluster_ids_ = [int(i) for i in ids]
        self.cluster_centers_ = centers
        self.inertia_ = float(inertia.value)
        return self

    def predict(self, X):
        """Index of the closest cluster centre for each sample in X
        (kcenters.py:104-126 -> libdistance.assign_nearest)."""
        X = _as_float(X)
        centers = self.cluster_centers_
        if is_device_array(centers):
            centers = centers.detach().cpu().numpy()
        xdt = np.dtype(str(X.dtype).replace("torch.", ""))
        if centers.dtype != xdt:
            raise TypeError('X and y must be both float32 or float64')
        labels, inertia = libdistance.assign_nearest(X, np.ascontiguousarray(centers), metric=self.metric)
        return labels

    def fit_predict(self, X, y=None):
        return self.fit(X, y).labels_


class KCenters(MultiSequenceClusterMixin, _KCenters, BaseEstimator):
    __doc__ = _KCenters.__doc__[: _KCenters.__doc__.find('Attributes')] + \
        '''
    Attributes
    ----------
    cluster_centers_ : (n_clusters, n_features)
    labels_ : list with one int array per input sequence (label of every frame)
    distances_ : list with one float64 array per input sequence (distance of every frame to its centre)
    '''

    def fit(self, sequences, y=None):
        """Fit the kcenters clustering on a list of [sequence_length, n_features] arrays."""
        MultiSequenceClusterMixin.fit(self, sequences)
        self.distances_ = self._split(self.distances_)
        return self

    def summarize(self):
        d = self.distances_
        if len(d) and is_device_array(d[0]):
            d = [x.detach().cpu().numpy() for x in d]
        return """KCenters clustering
--------------------
n_clusters : {n_clusters}
metric     : {metric}

Inertia       : {inertia}
Mean distance : {mean_distance}
Max  distance : {max_distance}
""".format(n_clusters=self.n_clusters, metric=self.metric,
           inertia=self.inertia_, mean_distance=np.mean(np.concatenate(d)),
           max_distance=np.max(np.concatenate(d)))
