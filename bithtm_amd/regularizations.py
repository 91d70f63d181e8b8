"""Parameter holders with the constructor signatures of bithtm/regularizations.py.

The computation they describe runs inside the HIP engine (kernels `k_sp_overlap`, `k_sel_pass`,
`k_sp_emit`); these objects carry the parameters and expose the state as attributes."""

import numpy as np


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

    def process(self, input_activation):
        raise NotImplementedError("boosting runs inside SpatialPooler.process on the GPU")

    update = process


class GlobalInhibition:
    """regularizations.py:24-29.  Selection is exact top-k with ties broken by lower column
    index; the winners are returned in ascending order."""

    def __init__(self, active_outputs):
        self.active_outputs = active_outputs

    def process(self, input_activation):
        raise NotImplementedError("inhibition runs inside SpatialPooler.process on the GPU")
