"""Decompositions of the MSMBuilder hot path on MI355X (reference: msmbuilder/decomposition)."""
from .tica import tICA

__all__ = ['tICA']
