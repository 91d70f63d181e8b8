"""bithtm_amd -- MI355X-native per-timestep hot path of bitHTM (Spatial Pooler + Temporal
Memory) behind the reference's Python class surface (bithtm/__init__.py:4-6).

    from bithtm_amd import HierarchicalTemporalMemory      # instead of: from bithtm import ...

The classes are host-side shells around hand-written HIP kernels for gfx950 reached through a
C ABI (include/bithtm_hip.h, libbithtm_hip.so).  There is no CPU fallback: importing the
engine without the built library raises ImportError."""

from . import networks
from .engine import CapacityError, HtmError  # noqa: F401
from .projections import DenseProjection, PredictiveProjection  # noqa: F401
from .regularizations import ExponentialBoosting, GlobalInhibition  # noqa: F401

SpatialPooler = networks.SpatialPooler
TemporalMemory = networks.TemporalMemory
HierarchicalTemporalMemory = networks.HierarchicalTemporalMemory
