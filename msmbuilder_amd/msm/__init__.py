"""Post-clustering counting step: ``_transition_counts`` as in ``msmbuilder.msm``."""
from .core import _transition_counts  # noqa: F401

__all__ = ['_transition_counts']
