"""List-of-sequences adaptor for clusterers (reference: msmbuilder/cluster/base.py:17-173).

Single-array estimators (``fit(X)`` / ``predict(X)`` on one [n, F] array) become
MSMBuilder estimators (``fit(list_of_sequences)``): the sequences are joined into one
array -- ``np.concatenate`` on the host, ``torch.cat`` for device-resident trajectories,
which stays in HBM -- the wrapped ``fit`` runs once, and ``labels_`` is cut back into
per-sequence pieces using the remembered lengths.  mdtraj trajectories (the RMSD
metric) are outside this package's scope.
"""
import numpy as np

from .._lib import is_device_array
from ..utils import check_iter_of_sequences

__all__ = ['MultiSequenceClusterMixin']


class MultiSequenceClusterMixin(object):
    _allow_trajectory = False

    def fit(self, sequences, y=None):
        """Cluster the frames of all sequences; ``labels_`` becomes a list with one
        integer array per input sequence."""
        check_iter_of_sequences(sequences, allow_trajectory=self._allow_trajectory)
        super(MultiSequenceClusterMixin, self).fit(self._concat(sequences))

        if hasattr(self, 'labels_'):
            self.labels_ = self._split(self.labels_)

        return self

    def _concat(self, sequences):
        self.__lengths = [len(s) for s in sequences]
        if len(sequences) > 0 and isinstance(sequences[0], np.ndarray):
            concat = np.ascontiguousarray(np.concatenate(sequences))
        elif len(sequences) > 0 and is_device_array(sequences[0]):
            import torch
            concat = torch.cat(list(sequences), dim=0).contiguous()
        else:
            raise TypeError('sequences must be a list of numpy arrays '
                            '(or torch CUDA tensors)')

        assert sum(self.__lengths) == len(concat)
        return concat

    def _split(self, concat):
        ends = np.cumsum(self.__lengths)
        return [concat[e - l: e] for (e, l) in zip(ends, self.__lengths)]

    def _split_indices(self, concat_inds):
        """Positions in the concatenated array -> (sequence index, frame index) pairs."""
        starts = np.append([0], np.cumsum(self.__lengths))
        table = np.zeros((starts[-1], 2), dtype=int)
        for traj_i, (a, b) in enumerate(zip(starts[:-1], starts[1:])):
            table[a:b, 0] = traj_i
            table[a:b, 1] = np.arange(b - a)
        return table[concat_inds]

    def predict(self, sequences, y=None):
        """Nearest-centre index for every frame of every sequence (a list of arrays)."""
        check_iter_of_sequences(sequences, allow_trajectory=self._allow_trajectory)
        return [self.partial_predict(X) for X in sequences]

    def partial_predict(self, X, y=None):
        """Nearest-centre index for every frame of one sequence."""
        return super(MultiSequenceClusterMixin, self).predict(X)

    def fit_predict(self, sequences, y=None):
        """``fit`` then return the training labels, one array per sequence."""
        if hasattr(super(MultiSequenceClusterMixin, self), 'fit_predict'):
            check_iter_of_sequences(sequences, allow_trajectory=self._allow_trajectory)
            labels = super(MultiSequenceClusterMixin, self).fit_predict(sequences)
        else:
            self.fit(sequences)
            labels = self.predict(sequences)

        if not isinstance(labels, list):
            labels = self._split(labels)
        return labels

    def transform(self, sequences):
        """Alias for predict"""
        return self.predict(sequences)

    def partial_transform(self, X):
        """Alias for partial_predict"""
        return self.partial_predict(X)

    def fit_transform(self, sequences, y=None):
        """Alias for fit_predict"""
        return self.fit_predict(sequences, y)
