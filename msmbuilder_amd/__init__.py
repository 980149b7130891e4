"""msmbuilder_amd -- MI355X (gfx950) implementation of MSMBuilder's tICA +
geometric-clustering hot path: ``decomposition.tICA``, ``cluster.KCenters``,
``cluster.MiniBatchKMeans`` and the ``libdistance`` module, as drop-ins for their
``msmbuilder`` namesakes.  All numerics run in hand-written HIP kernels
(msmbuilder_amd/csrc -> libmsmhip.so) behind the C ABI of include/msmhip.h;
there is no CPU fallback."""
__version__ = "0.1.0"

from . import libdistance  # noqa: F401
from . import preprocessing  # noqa: F401
from .cluster import KCenters, MiniBatchKMeans  # noqa: F401
from .decomposition import tICA  # noqa: F401
