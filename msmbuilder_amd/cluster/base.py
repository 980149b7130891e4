"""List-of-sequences adaptor for clusterers (reference behaviour: msmbuilder/cluster/base.py:17-173).

MSMBuilder estimators take a LIST of trajectories; the clustering kernels take one [n, F] array.  This mixin sits
between the two: it joins the trajectories (``torch.cat`` for device-resident ones, so they never leave HBM;
``numpy.concatenate`` for host arrays), lets the wrapped single-array estimator run once, and cuts ``labels_`` back
into one piece per trajectory with the remembered lengths.  ``predict`` / ``transform`` work trajectory by trajectory (in one launch when the trajectories are views of one
allocation).
mdtraj trajectories (RMSD metric) are outside this package's scope.
"""
import numpy as np

from .._lib import is_device_array
from ..utils import check_iter_of_sequences

__all__ = ['MultiSequenceClusterMixin']


def _join(sequences):
    """One array holding every frame, on the side (host / device) the first trajectory lives on."""
    if not len(sequences):
        raise TypeError('sequences must be a list of numpy arrays (or torch CUDA tensors)')
    head = sequences[0]
    if len(sequences) == 1:   # one trajectory: itself (torch.cat / np.concatenate of one piece is a full copy: 0.3 ms per 0.8 GB)
        if isinstance(head, np.ndarray):
            return np.ascontiguousarray(head)
        if is_device_array(head):
            return head.contiguous()
    from .._lib import adjacent_view
    joined = adjacent_view(sequences) if isinstance(sequences, (list, tuple)) else None
    if joined is not None:
        return joined   # the trajectories already lie back to back in one allocation: no copy
    if isinstance(head, np.ndarray):
        return np.ascontiguousarray(np.concatenate(sequences))
    if is_device_array(head):
        import torch
        return torch.cat(list(sequences), dim=0).contiguous()
    raise TypeError('sequences must be a list of numpy arrays (or torch CUDA tensors)')


class MultiSequenceClusterMixin(object):
    _allow_trajectory = False

    # ------------------------------------------------------------------ bookkeeping
    def _concat(self, sequences):
        self._seq_lengths = [len(s) for s in sequences]
        joined = _join(sequences)
        if len(joined) != sum(self._seq_lengths):
            raise ValueError('sequences must be a list of sequences')
        return joined

    def _split(self, joined):
        """Per-trajectory views of an array indexed like the joined frames."""
        out, start = [], 0
        for n in self._seq_lengths:
            out.append(joined[start:start + n])
            start += n
        return out

    def _split_indices(self, joined_positions):
        """(trajectory index, frame index) for positions in the joined array."""
        bounds = np.concatenate(([0], np.cumsum(self._seq_lengths)))
        pos = np.asarray(joined_positions)
        traj = np.searchsorted(bounds, pos, side='right') - 1
        return np.stack([traj, pos - bounds[traj]], axis=-1).astype(int)

    # ------------------------------------------------------------------ estimator protocol
    def fit(self, sequences, y=None):
        """Cluster the frames of all trajectories; ``labels_`` becomes a list with one integer array each."""
        check_iter_of_sequences(sequences, allow_trajectory=self._allow_trajectory)
        super(MultiSequenceClusterMixin, self).fit(self._concat(sequences))
        if hasattr(self, 'labels_'):
            self.labels_ = self._split(self.labels_)
        return self

    def partial_predict(self, X, y=None):
        """Nearest-centre index for every frame of ONE trajectory."""
        return super(MultiSequenceClusterMixin, self).predict(X)

    def predict(self, sequences, y=None):
        """Nearest-centre index for every frame of every trajectory (a list of arrays)."""
        check_iter_of_sequences(sequences, allow_trajectory=self._allow_trajectory)
        # trajectories that lie back to back in one allocation (views of a joined array / tensor) are labelled in ONE
        # launch and the labels cut per trajectory: a launch per 10,000-frame trajectory fills a sixth of the GPU
        from .._lib import adjacent_view, cut_rows
        joined = adjacent_view(sequences) if isinstance(sequences, (list, tuple)) else None
        if joined is not None:
            return cut_rows(self.partial_predict(joined), [X.shape[0] for X in sequences])
        # separately allocated device trajectories of a few columns (a tICA projection): joining them costs less than a
        # launch per trajectory (10M x 10 float64 = 0.8 GB = 0.4 ms of copy against a thousand launches)
        if (isinstance(sequences, (list, tuple)) and len(sequences) > 1 and all(is_device_array(X) for X in sequences)):
            try:
                total = sum(X.numel() * X.element_size() for X in sequences)
                same = all(X.dim() == 2 and X.shape[1] == sequences[0].shape[1] and X.dtype == sequences[0].dtype
                           and X.device == sequences[0].device for X in sequences)
            except Exception:
                same, total = False, 0
            if same and total <= (1 << 30):
                import torch
                from .._lib import cut_rows
                return cut_rows(self.partial_predict(torch.cat(list(sequences), dim=0)), [X.shape[0] for X in sequences])
        return [self.partial_predict(X) for X in sequences]

    def fit_predict(self, sequences, y=None):
        """``fit``, then the training labels, one array per trajectory."""
        wrapped = super(MultiSequenceClusterMixin, self)
        if hasattr(wrapped, 'fit_predict'):
            check_iter_of_sequences(sequences, allow_trajectory=self._allow_trajectory)
            labels = wrapped.fit_predict(sequences)
        else:
            labels = self.fit(sequences).predict(sequences)
        return labels if isinstance(labels, list) else self._split(labels)

    # the reference exposes the same three operations under transformer names
    def transform(self, sequences):
        return self.predict(sequences)

    def partial_transform(self, X):
        return self.partial_predict(X)

    def fit_transform(self, sequences, y=None):
        return self.fit_predict(sequences, y)
