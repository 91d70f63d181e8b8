"""Counterparts of bithtm/regularizations.py with the same constructor signatures.

Inside a SpatialPooler the computation they describe runs in the HIP engine's fused timestep (kernels
`k_sp_overlap`, `k_sel_pass`, `k_sp_emit`) and these objects carry the parameters and expose the state.  Called on
their own -- `process(...)` / `update(...)` as in the reference -- they run the same device kernels one phase at a
time (htm_sp_phase, include/bithtm_hip.h); an object that is not part of a SpatialPooler creates a small engine of
its own for that on first use."""

import numpy as np

from . import _lib as L


class _Placeholder:
    """Parameters of a DenseProjection nobody evaluates (an engine needs some Spatial Pooler storage)."""

    def __init__(self, input_dim, output_dim):
        self.input_dim, self.output_dim = input_dim, output_dim
        self.permanence_threshold, self.permanence_increment, self.permanence_decrement = 0.0, 0.03, 0.015
        self._engine = None
        self._permanence = np.zeros((output_dim, input_dim), dtype=np.float64)


class ExponentialBoosting:
    """regularizations.py:4-21.  `duty_cycle` reads the float32 duty cycle from the device."""

    def __init__(self, output_dim, active_outputs, intensity=0.3, momentum=0.99):
        self.output_dim = output_dim
        self.active_outputs = active_outputs
        self.density = active_outputs / output_dim
        self.intensity = intensity
        self.momentum = momentum
        self._engine = None
        self._duty_cycle = np.zeros(output_dim, dtype=np.float32)

    @property
    def duty_cycle(self):
        if self._engine is not None:
            return self._engine.read_duty_cycle()
        return self._duty_cycle

    def _ensure_engine(self):
        if self._engine is None:
            from .engine import Engine
            eng = Engine(32, self.output_dim, 0, self.active_outputs, proximal=_Placeholder(32, self.output_dim), boosting=self)
            eng.write(L.F_DUTY_CYCLE, self._duty_cycle, np.float32)
            self._engine = eng
        return self._engine

    def process(self, input_activation):
        """regularizations.py:15-17 on the device: float32 factor (the documented exp), exact float64 product."""
        eng = self._ensure_engine()
        eng.sp_phase(L.SP_BOOST, np.asarray(input_activation), np.int32)
        return eng.read(L.F_BOOSTED, np.float64, self.output_dim)

    def update(self, active_input):
        """regularizations.py:19-21 on the device."""
        eng = self._ensure_engine()
        eng.sp_phase(L.SP_ACTIVE, np.asarray(active_input), np.int32)
        eng.sp_phase(L.SP_DUTY)


class GlobalInhibition:
    """regularizations.py:24-29.  Selection is exact top-k with ties broken by lower column
    index; the winners are returned in ascending order."""

    def __init__(self, active_outputs):
        self.active_outputs = active_outputs
        self._engine = None

    def process(self, input_activation):
        """regularizations.py:28-29 on the device (radix select of the k largest, ties to the lower index)."""
        x = np.ascontiguousarray(input_activation, dtype=np.float64)
        if (x < 0).any():
            raise ValueError("GlobalInhibition.process: boosted overlaps are non-negative")
        eng = self._engine
        if eng is None or eng.column_dim != len(x):
            from .engine import Engine
            eng = self._engine = Engine(32, len(x), 0, self.active_outputs, proximal=_Placeholder(32, len(x)),
                                        boosting=ExponentialBoosting(len(x), self.active_outputs))
        eng.sp_phase(L.SP_SELECT, x, np.float64)
        return eng.read(L.F_ACTIVE_COLUMN, np.int32, eng.active_columns).astype(np.int64)
