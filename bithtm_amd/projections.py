"""Parameter / initial-state holders with the constructor signatures of bithtm/projections.py.

The projections themselves live in device memory and are evaluated by the HIP engine; these
objects carry the parameters (so non-default values are passed exactly as with the reference:
`SpatialPooler(..., proximal_projection=DenseProjection(I, C, permanence_threshold=0.02))`) and
give attribute access to the state."""

import numpy as np


class DenseProjection:
    """projections.py:6-24.  The constructor draws the initial permanences from the global
    NumPy RNG with the reference's expression (projections.py:16), so a seeded script gets the
    same matrix."""

    def __init__(self, input_dim, output_dim, permanence_mean=0.0, permanence_std=0.1,
                 permanence_threshold=0.0, permanence_increment=0.03, permanence_decrement=0.015):
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.permanence_threshold = permanence_threshold
        self.permanence_increment = permanence_increment
        self.permanence_decrement = permanence_decrement
        self._engine = None
        self._permanence = np.random.randn(output_dim, input_dim) * permanence_std + permanence_mean

    @property
    def permanence(self):
        if self._engine is not None:
            return self._engine.get_permanence()
        return self._permanence

    @permanence.setter
    def permanence(self, value):
        value = np.ascontiguousarray(value, dtype=np.float64)
        assert value.shape == (self.output_dim, self.input_dim)
        if self._engine is not None:
            self._engine.set_permanence(value)
        else:
            self._permanence = value

    def process(self, input_activation):
        raise NotImplementedError("the overlap runs inside SpatialPooler.process on the GPU")

    update = process


class PredictiveProjection:
    """projections.py:194-293 (with the SparseProjection store of :27-192 behind it).

    Extra keyword arguments size the fixed-capacity device pool that replaces the reference's
    growing arrays (utils.py:79-135): `segment_capacity` segments of `segment_slots` synapse
    slots each.  Running out of either raises CapacityError; nothing is dropped silently."""

    def __init__(self, output_dim, permanence_initial=0.21, permanence_threshold=0.5, permanence_increment=0.1,
                 permanence_decrement=0.1, permanence_punishment=0.01, segment_activation_threshold=15,
                 segment_matching_threshold=15, segment_sampling_synapses=32,
                 segment_bundle_growth_exponential=True, segment_capacity=None, segment_slots=128,
                 segment_capacity_local=None):
        assert segment_activation_threshold >= segment_matching_threshold      # projections.py:211
        self.output_dim = output_dim
        self.permanence_initial = permanence_initial
        self.permanence_threshold = permanence_threshold
        self.permanence_increment = permanence_increment
        self.permanence_decrement = permanence_decrement
        self.permanence_punishment = permanence_punishment
        self.segment_activation_threshold = segment_activation_threshold
        self.segment_matching_threshold = segment_matching_threshold
        self.segment_sampling_synapses = segment_sampling_synapses
        self.segment_capacity = segment_capacity
        self.segment_slots = segment_slots
        self.segment_capacity_local = segment_capacity_local      # column-sharded engines: rows for the rank's own segments
        self._engine = None

    # state views (copy_custom-style consumers: reference_implementations.py:51-66)
    @property
    def segment_bundle(self):
        return self._engine.read_store()["seg_cell"][:, None]

    @property
    def bundle_segments(self):
        return self._engine.read_store()["segcount"]

    def process(self, active_input, return_jittered_potential_info=True):
        raise NotImplementedError("the segment scan runs inside TemporalMemory.process on the GPU")

    update = process
