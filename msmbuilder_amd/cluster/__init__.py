"""Clusterers of the MSMBuilder hot path on MI355X (reference: msmbuilder/cluster/__init__.py)."""
from .base import MultiSequenceClusterMixin
from .kcenters import KCenters
from .minibatchkmeans import MiniBatchKMeans

__all__ = ['KCenters', 'MiniBatchKMeans', 'MultiSequenceClusterMixin']
