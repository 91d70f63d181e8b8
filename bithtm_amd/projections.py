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

    def _ensure_engine(self):
        if self._engine is None:
            from .engine import Engine
            from .regularizations import ExponentialBoosting
            # an engine of its own: a DenseProjection used outside a SpatialPooler (k is a formality here)
            k = min(self.output_dim, 2048)
            eng = Engine(self.input_dim, self.output_dim, 0, k, proximal=self, boosting=ExponentialBoosting(self.output_dim, k))
            self._engine, self._permanence = eng, None
        return self._engine

    def process(self, input_activation):
        """projections.py:18-21 on the device: popcount of (connected mask & input) per row."""
        from . import _lib as L
        eng = self._ensure_engine()
        eng.sp_phase(L.SP_OVERLAP, np.asarray(input_activation, dtype=np.bool_))
        return eng.read(L.F_OVERLAPS, np.int32, self.output_dim).astype(np.int64)

    def update(self, input_activation, learning_output):
        """projections.py:23-24 on the device (float64 rows, fused with the rebuild of their connected mask)."""
        from . import _lib as L
        eng = self._ensure_engine()
        cols = np.unique(np.asarray(learning_output, dtype=np.int64).reshape(-1))      # (fancy-indexed += applies once per distinct row)
        chunk = max(eng.active_columns, 1)
        for lo in range(0, len(cols), chunk):
            eng.sp_phase(L.SP_ACTIVE, cols[lo:lo + chunk], np.int32)
            eng.sp_phase(L.SP_LEARN, np.asarray(input_activation, dtype=np.bool_))


class SegmentProjectionView:
    """The read surface of the reference's SparseProjection (projections.py:27-68) over a snapshot of the device's
    segment store: what `reference_implementations.TemporalMemory.copy_custom` (:48-88) and other consumers of the
    reference's layout touch.

      output_edges[S, 1]       valid synapses per segment                       (projections.py:42)
      output_edge[S, E]        slot_in_input_edge * (input_dim + 1) + target    (:43, :63-64); invalid = input_dim (:36)
      output_permanence[S, E]  float32, -1.0 where invalid                      (:44, :58)
      input_edge[N + 1, E_in]  per presynaptic cell the list of 1 + segment, 0 = free (:40, :35); row N is the pad row

    The device keeps the pull form only (one packed row of presynaptic ids per segment); the mirrored push form is
    rebuilt here, on the host, when somebody asks for it: slots of a cell's input_edge row are handed out in segment
    order."""

    def __init__(self, presyn, perm, input_dim):
        presyn = np.asarray(presyn)
        S, E = presyn.shape
        N = int(input_dim)
        self.input_dim, self.output_dim = N, S
        self.invalid_input_edge, self.invalid_output_edge = 0, N
        valid = presyn >= 0
        seg, slot = np.nonzero(valid)                         # row-major: ascending segment, then slot
        target = presyn[seg, slot].astype(np.int64)
        order = np.argsort(target, kind="stable")             # groups by presynaptic cell, segments ascending inside
        t_sorted = target[order]
        first = np.flatnonzero(np.r_[True, t_sorted[1:] != t_sorted[:-1]]) if len(order) else np.zeros(0, np.int64)
        sizes = np.diff(np.r_[first, len(order)])
        in_slot = np.arange(len(order)) - np.repeat(first, sizes)
        e_in = int(sizes.max(initial=0))
        packed = np.empty(len(order), dtype=np.int64)
        packed[order] = in_slot * (N + 1) + t_sorted
        dtype = np.int32 if (e_in * (N + 1) + N) < 2 ** 31 else np.int64     # (the reference's int32 packing would overflow)
        self.output_edge = np.full((S, E), N, dtype=dtype)
        self.output_edge[seg, slot] = packed
        self.output_permanence = np.where(valid, np.asarray(perm, dtype=np.float32), np.float32(-1.0))
        self.output_edges = valid.sum(axis=1, dtype=np.int32)[:, None]
        self.input_edge = np.zeros((N + 1, e_in), dtype=np.int32)
        self.input_edge[t_sorted, in_slot] = 1 + seg[order]

    def get_output_edge_target(self, output_edge):            # projections.py:60-61
        return output_edge % (self.input_dim + 1)

    def pack_output_edge(self, target_input, input_edge):     # :63-64
        return input_edge * (self.input_dim + 1) + target_input

    def unpack_output_edge(self, output_edge):                # :66-68
        input_edge, target_input = np.divmod(output_edge, self.input_dim + 1)
        return target_input, input_edge


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
        from . import _lib as L
        eng = self._ensure_engine()
        return eng.read(L.F_SEG_CELL, np.int32, eng.info().local_segments)[:, None]

    @property
    def bundle_segments(self):
        from . import _lib as L
        eng = self._ensure_engine()
        return eng.read(L.F_SEGCOUNT, np.int32, eng.column_dim * eng.cell_dim)[:self.output_dim]

    @property
    def segment_projection(self):
        """A snapshot of the synapse store in the reference's SparseProjection layout (see SegmentProjectionView):
        `reference_implementations.TemporalMemory().copy_custom(tm)` accepts a bithtm_amd TemporalMemory through it."""
        st = self._ensure_engine().read_store()
        return SegmentProjectionView(st["presyn"], st["perm"], self.output_dim)

    # ---- the reference's methods, callable on their own (projections.py:229-293): a caller's own TemporalMemory around
    # the device's segment store.  Inside bithtm_amd.TemporalMemory they are not called: its timestep is fused on the device.
    class State:
        """projections.py:195-203."""

        def __init__(self, d, jitter=True):
            self.prediction = d["prediction"]
            self.segment_potential = d["segment_potential"]
            self.matching_segment = d["matching_segment"]
            self.matching_segment_activation = d["matching_segment_activation"]
            self.matching_segment_active = d["matching_segment_active"]
            self.max_jittered_potential = d["max_jittered_potential"] if jitter else None
            self.matching_segment_jittered_potential = d["matching_segment_jittered_potential"] if jitter else None
            self._jitter = (d["max_jittered_potential"], d["matching_segment_jittered_potential"])

    def _ensure_engine(self):
        """An engine of its own for a projection used outside a fused TemporalMemory.  The projection lives in CELL space
        (segments belong to cells, synapses point at cells; what a column is matters to TemporalMemory.process only), so
        any model shape fits the device's layout of 32 cells per word: `cell_dim` <= 32 dividing output_dim keeps the
        model's own columns, anything else -- 33 cells per column and more -- is laid out as ceil(output_dim / 32) words
        of 32 cells, flat cell ids unchanged."""
        if self._engine is None:
            from .engine import Engine
            cell_dim = int(getattr(self, "cell_dim", None) or 32)
            if cell_dim > 64 or self.output_dim % cell_dim:
                cell_dim = 32
            C = -(-self.output_dim // cell_dim)
            self._engine = Engine(0, C, cell_dim, int(getattr(self, "active_columns", None) or C), distal=self,
                                  seed=int(getattr(self, "seed", 0)))
            self._last_state = None
        return self._engine

    def _padded(self, a, fill=0):
        """A per-cell array of the model (output_dim entries) in the engine's length (whole words of cells)."""
        eng = self._ensure_engine()
        a = np.asarray(a).reshape(-1)
        n = eng.column_dim * eng.cell_dim
        if len(a) == n:
            return a
        out = np.full(n, fill, dtype=a.dtype)
        out[:len(a)] = a
        return out

    def fill_jittered_potential_info(self, state, matching_segment_bundle=None):
        """projections.py:229-239 (the scan computed both with the keyed draws)."""
        if state.max_jittered_potential is None or state.matching_segment_jittered_potential is None:
            state.max_jittered_potential, state.matching_segment_jittered_potential = state._jitter

    def get_jittered_potential_info(self, state, matching_segment_bundle=None):
        """projections.py:241-243."""
        self.fill_jittered_potential_info(state)
        return state.max_jittered_potential, state.matching_segment_jittered_potential

    def process(self, active_input, return_jittered_potential_info=True):
        """projections.py:245-255 on the device (the segment scan): `active_input` = flat ids of the active cells."""
        from .engine import CapacityError  # noqa: F401
        eng = self._ensure_engine()
        K, C = eng.cell_dim, eng.column_dim
        flat = np.asarray(active_input, dtype=np.int64).reshape(-1)
        wpc = eng.cell_words // C                    # (words per column: the cell j of a column is bit j % 32 of its word j // 32)
        words = np.zeros(C * wpc, dtype=np.uint32)
        np.bitwise_or.at(words, (flat // K) * wpc + (flat % K) // 32, (np.uint32(1) << ((flat % K) % 32).astype(np.uint32)))
        eng.tm_scan(words)
        eng.check_capacity()
        d = eng.read_distal()
        d["prediction"], d["max_jittered_potential"] = d["prediction"][:self.output_dim], d["max_jittered_potential"][:self.output_dim]
        st = PredictiveProjection.State(d, jitter=return_jittered_potential_info)
        self._last_state = st
        return st

    def update(self, prev_state, input_activation, learning_output, output_punishment, winner_input=None, output_learning=None,
               epsilon=1e-8):
        """projections.py:257-293 on the device (segment allocation, learn / punish classification, permanence updates,
        growth).  `prev_state`: the State this object's process() returned last -- or any earlier one, which is then
        written back as the device's previous step (a host round trip, like TemporalMemory.process(prev_state=))."""
        if prev_state is None:                                                    # :258-259
            return
        eng = self._ensure_engine()
        K, C = eng.cell_dim, eng.column_dim
        if getattr(eng, "_epsilon", 1e-8) != epsilon:
            eng.set_epsilon(epsilon)
            eng._epsilon = epsilon
        self.fill_jittered_potential_info(prev_state)
        learning_output = np.asarray(learning_output, dtype=np.int64).reshape(-1)
        input_activation = self._padded(np.asarray(input_activation, dtype=np.bool_)).reshape(C, K)
        # the previous step's side of the call: prev_state, input_activation, winner_input
        from types import SimpleNamespace
        prev = SimpleNamespace(matching_segment=prev_state.matching_segment, segment_potential=prev_state.segment_potential,
                               matching_segment_activation=prev_state.matching_segment_activation,
                               matching_segment_active=prev_state.matching_segment_active,
                               matching_segment_jittered_potential=prev_state.matching_segment_jittered_potential,
                               max_jittered_potential=self._padded(np.asarray(prev_state.max_jittered_potential, dtype=np.float32)))
        eng.import_prev_state(self._padded(np.asarray(prev_state.prediction)).reshape(C, K) > epsilon, input_activation,
                              None if winner_input is None else np.asarray(winner_input, dtype=np.int64), prev)
        learn_mask = np.zeros(C * K, dtype=np.bool_)
        if output_learning is None:
            learn_mask[learning_output] = True                                    # :261-262
        else:
            learn_mask[:] = self._padded(np.asarray(output_learning, dtype=np.bool_))
        unacc = learning_output[np.asarray(prev_state.max_jittered_potential)[learning_output] < np.float32(epsilon)]      # :271
        need = np.zeros(C * K, dtype=np.bool_)
        need[unacc] = True
        learn_mask |= need                          # (a cell that gets a segment is a learning cell: :281 learns on the new segment)
        from .engine import bool_to_words
        wpc = eng.cell_words // C
        ww, uw = bool_to_words(learn_mask.reshape(C, K)).reshape(C, wpc), bool_to_words(need.reshape(C, K)).reshape(C, wpc)
        cols = np.flatnonzero(ww.any(axis=1))
        # output_punishment=None: the mask TemporalMemory builds (networks.py:107-108,111) -- every cell of a column without a
        # learning cell -- is built by the library (htm_tm_update's punish_words == NULL)
        eng.tm_update(cols, ww[cols], uw[cols], None if output_punishment is None else
                      bool_to_words(self._padded(np.asarray(output_punishment, dtype=np.bool_)).reshape(C, K)))
