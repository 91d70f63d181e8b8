"""Drop-in counterparts of bithtm/networks.py: SpatialPooler, TemporalMemory,
HierarchicalTemporalMemory with the reference's constructor arguments, `process(...)`
(plus the alias `compute`), `State` records and `last_state`.

All computation happens in the HIP engine (bithtm_amd/csrc/htm_engine.hip) behind the C ABI;
this module is host logic: parameter plumbing, lazy read-back of State fields, and the
NumPy dtypes / shapes the reference returns.

Differences from the reference, all documented in DESIGN.md:
  * `active_column` comes back in ascending order (reference: np.argpartition order) and ties at
    the k-th boosted overlap go to the lower column index;
  * random tie-breaks use keyed draws (`seed=`), not the global np.random stream;
  * plug-in arguments (`proximal_projection=`, `boosting=`, `inhibition=`: networks.py:16,22-24) accept the objects
    of bithtm_amd.projections / bithtm_amd.regularizations (same constructor signatures as the reference's) -- the
    timestep is then one fused device call -- and ANY other object with the reference's `process` / `update`
    methods: SpatialPooler.process then runs phase by phase (htm_sp_phase), the device doing the parts that are
    its own and the user's object being called on the host for the others, exactly in the order of networks.py:26-35.
    `distal_projection=` (networks.py:50,55) likewise: a bithtm_amd PredictiveProjection is the device's segment store and the
    timestep stays fused; any other object with the reference's PredictiveProjection interface is called on the host from a
    host-side TemporalMemory.process (networks.py:91-128, `_process_host`).  `spatial_pooler=` / `temporal_memory=`
    (networks.py:134,143-144; example.py:7-12 swaps the Temporal Memory) take any object with `process`.
"""

import weakref

import numpy as np

from . import _lib as L
from .engine import Engine, words_to_bool
from .projections import DenseProjection, PredictiveProjection
from .regularizations import ExponentialBoosting, GlobalInhibition, _Placeholder


class _Lazy:
    """State base: fields are fetched from the device on first access while the engine still
    holds that step; the engine materialises every live State before it overwrites the step."""

    _fields = ()

    def __init__(self, engine, step):
        self._engine = engine
        self._step = step
        self._cache = {}
        engine_states(engine).append(weakref.ref(self))

    def _fetch(self):
        raise NotImplementedError

    def _materialize(self):
        if self._engine is not None:
            if self._engine.steps != self._step:
                raise RuntimeError("State outlived its timestep without being materialised")
            self._cache.update(self._fetch())
            self._engine = None

    def __getattr__(self, name):
        if name.startswith("_") or name not in type(self)._fields:
            raise AttributeError(name)
        if name not in self._cache:
            self._materialize()
        return self._cache[name]

    def __setattr__(self, name, value):
        if name.startswith("_"):
            object.__setattr__(self, name, value)
        else:
            self._cache[name] = value


def engine_states(engine):
    """Weak references to the States of the engine's current step (a plain list: a host-fed step creates two and retires
    two, and this is on its path)."""
    s = getattr(engine, "_live_states", None)
    if s is None:
        s = engine._live_states = []
    return s


def retire_states(engine):
    """Called before a new step is enqueued: pin down every State somebody still holds."""
    live = getattr(engine, "_live_states", None)
    if live:
        for ref in live:
            st = ref()
            if st is not None:
                st._materialize()
        live.clear()


class SpatialPooler:
    class State(_Lazy):
        """networks.py:8-12."""
        _fields = ("active_column", "overlaps", "boosted_overlaps")

        def _fetch(self):
            return self._engine.read_sp_fields()

        @classmethod
        def _eager(cls, active_column, overlaps, boosted_overlaps):
            st = cls.__new__(cls)
            object.__setattr__(st, "_engine", None)
            object.__setattr__(st, "_step", -1)
            object.__setattr__(st, "_cache", dict(active_column=active_column, overlaps=overlaps, boosted_overlaps=boosted_overlaps))
            return st

    def __init__(self, input_dim, column_dim, active_columns, proximal_projection=None, boosting=None,
                 inhibition=None, device=0):
        self.input_dim = input_dim
        self.column_dim = column_dim
        self.active_columns = active_columns
        self.device = device
        self.proximal_projection = proximal_projection or DenseProjection(input_dim, column_dim)       # networks.py:22
        self.boosting = boosting or ExponentialBoosting(column_dim, active_columns)                    # :23
        self.inhibition = inhibition or GlobalInhibition(active_columns)                               # :24
        for name, obj, methods in (("proximal_projection", self.proximal_projection, ("process", "update")),
                                   ("boosting", self.boosting, ("process", "update")), ("inhibition", self.inhibition, ("process",))):
            if not all(callable(getattr(obj, m, None)) for m in methods):
                raise TypeError(f"{name} must have the reference's {' / '.join(methods)} methods; got {type(obj).__name__}")
        # which parts are the device's own (exact types: a subclass may override the methods the device would skip)
        self._own_proximal = type(self.proximal_projection) is DenseProjection
        self._own_boosting = type(self.boosting) is ExponentialBoosting
        self._own_inhibition = type(self.inhibition) is GlobalInhibition
        self._engine = None
        self._fused = False

    @property
    def _plain(self):
        return self._own_proximal and self._own_boosting and self._own_inhibition

    def _engine_parts(self):
        """The parameter objects the engine is built from: the user's where they are the device's own kind."""
        proximal = self.proximal_projection if self._own_proximal else _Placeholder(self.input_dim, self.column_dim)
        boosting = self.boosting if self._own_boosting else ExponentialBoosting(self.column_dim, self.active_columns)
        return proximal, boosting

    def _bind(self, engine, fused):
        self._engine, self._fused = engine, fused
        if self._own_proximal:
            self.proximal_projection._engine = engine
            self.proximal_projection._permanence = None      # now lives in device memory
        if self._own_boosting:
            self.boosting._engine = engine

    def _ensure_engine(self):
        if self._engine is None:
            proximal, boosting = self._engine_parts()
            self._bind(Engine(self.input_dim, self.column_dim, 0, self.active_columns, proximal=proximal, boosting=boosting,
                              device=self.device), False)
        return self._engine

    def _process_phases(self, input, learning, commit):
        """networks.py:26-35 with some of the three objects living on the host: the device runs the phases that are its
        own (htm_sp_phase), the user's objects are called for the others, in the reference's order."""
        eng = self._engine
        C = self.column_dim
        input = np.asarray(input, dtype=np.bool_)
        overlaps = boosted = None
        if self._own_proximal:
            eng.sp_phase(L.SP_OVERLAP, input)                                         # :27 (and :28 with the device's boosting)
        else:
            overlaps = np.asarray(self.proximal_projection.process(input))
            if self._own_boosting:
                eng.sp_phase(L.SP_BOOST, overlaps, np.int32)                          # :28
        if not self._own_boosting:
            if overlaps is None:
                overlaps = eng.read(L.F_OVERLAPS, np.int32, C).astype(np.int64)
            boosted = np.asarray(self.boosting.process(overlaps))                     # :28
        if self._own_inhibition:
            eng.sp_phase(L.SP_SELECT, None if self._own_boosting else boosted, np.float64)     # :29
            active_column = eng.read(L.F_ACTIVE_COLUMN, np.int32, self.active_columns).astype(np.int64)
        else:
            if boosted is None:
                boosted = eng.read(L.F_BOOSTED, np.float64, C)
            active_column = np.asarray(self.inhibition.process(boosted))              # :29
            eng.sp_phase(L.SP_ACTIVE, active_column, np.int32)
        if learning:                                                                  # :31-32
            if self._own_proximal:
                eng.sp_phase(L.SP_LEARN)
            else:
                self.proximal_projection.update(input, active_column)
        if self._own_boosting:                                                        # :33
            eng.sp_phase(L.SP_DUTY)
        else:
            self.boosting.update(active_column)
        if overlaps is None:
            overlaps = eng.read(L.F_OVERLAPS, np.int32, C).astype(np.int64)
        if boosted is None:
            boosted = eng.read(L.F_BOOSTED, np.float64, C)
        if commit:
            eng.sp_phase(L.SP_COMMIT)
        return self.State._eager(active_column, overlaps, boosted)

    def process(self, input, learning=True):
        """networks.py:26-35."""
        eng = self._ensure_engine()
        if self._fused:
            raise RuntimeError("this SpatialPooler is fused into a HierarchicalTemporalMemory; call its process()")
        retire_states(eng)
        if not self._plain:
            return self._process_phases(input, learning, commit=True)
        eng.sp_step(input, learning=learning)
        return self.State(eng, eng.steps)

    compute = process


class TemporalMemory:
    class State(_Lazy):
        """networks.py:39-46; `distal_state` mirrors PredictiveProjection.State (projections.py:195-203)."""
        _fields = ("active_cell", "winner_cell", "cell_activation", "cell_prediction", "active_column_bursting",
                   "distal_state")

        def __init__(self, engine, step, active_column=None):
            self._active_column = active_column
            super().__init__(engine, step)

        def _fetch(self):
            eng = self._engine
            C, K = eng.column_dim, eng.cell_dim
            info = eng.check_capacity()          # an overflowed pool must not be read as if nothing happened
            cols = self._active_column
            if cols is None:
                cols = eng.read(L.F_ACTIVE_COLUMN, np.int32, eng.active_columns).astype(np.int64)
            cols = np.asarray(cols, dtype=np.int64)             # in the CALLER's order, as the reference indexes with it
            act = words_to_bool(eng.read(L.F_CELL_ACTIVATION, np.uint32, eng.cell_words), K)
            rows, cells = np.where(act[cols])                                        # networks.py:116-117
            bursting = np.empty(len(cols), dtype=np.bool_)      # (the device keeps it by ascending column)
            bursting[np.argsort(cols, kind="stable")] = eng.read(L.F_BURSTING, np.uint8, eng.active_columns)[:len(cols)].astype(np.bool_)
            out = dict(
                cell_activation=act,
                cell_prediction=words_to_bool(eng.read(L.F_CELL_PREDICTION, np.uint32, eng.cell_words), K),
                active_cell=(cols[rows], cells),
                active_column_bursting=bursting[:, None],
                winner_cell=None)
            if info.has_winner_cells:
                winner = words_to_bool(eng.read(L.F_WINNER_WORDS, np.uint32, eng.cell_words), K)
                rows, cells = np.where(winner[cols])                                 # networks.py:103-104
                out["winner_cell"] = (cols[rows], cells)
            d = eng.read_distal()
            out["distal_state"] = None if d is None else _DistalState(d)
            return out

    def __init__(self, column_dim, cell_dim, distal_projection=None, seed=0, device=0):
        self.column_dim = column_dim
        self.cell_dim = cell_dim
        self.seed = seed
        self.device = device
        self.distal_projection = distal_projection or PredictiveProjection(self.column_dim * self.cell_dim)     # networks.py:55
        # the device's own kind (exact type: a subclass may override the methods the fused step would skip) -- or any object
        # with the reference's PredictiveProjection interface, which is then called on the host (_process_host)
        # (more than 64 cells per column: the device's Temporal Memory step is built on one or two 32-bit words of cells per
        # column -- a lane per cell, a half-wave or a wave per column; the segment store is not -- it lives in cell space -- so
        # such a model keeps its projection on the device and runs the per-column part of TemporalMemory.process,
        # networks.py:95-119, on the host: correct, not fast)
        self._own_distal = type(self.distal_projection) is PredictiveProjection and cell_dim <= 64
        if not self._own_distal:
            if isinstance(self.distal_projection, PredictiveProjection):      # a subclass of the device's: tell it the model's shape
                self.distal_projection.cell_dim, self.distal_projection.seed = cell_dim, seed
            missing = [m for m in ("process", "update", "get_jittered_potential_info") if not callable(getattr(self.distal_projection, m, None))]
            missing += [a for a in ("segment_matching_threshold", "bundle_segments") if a not in dir(self.distal_projection)]
            if missing:
                raise TypeError(f"distal_projection must have the reference's PredictiveProjection interface; "
                                f"{type(self.distal_projection).__name__} lacks {', '.join(missing)}")
        self._host_step = 0
        self._host_last = None
        self._engine = None
        self._fused = False
        self._last_ref = None             # weak: an unread State costs nothing
        self._last_cols = None
        self._empty_state = self.get_empty_state()                                   # networks.py:57

    def _bind(self, engine, fused):
        self._engine, self._fused = engine, fused
        self.distal_projection._engine = engine

    def _ensure_engine(self, n_active):
        if self._engine is None:
            self._bind(Engine(0, self.column_dim, self.cell_dim, n_active, distal=self.distal_projection,
                              seed=self.seed, device=self.device), False)
        return self._engine

    def grow_pool(self, segment_capacity=None, segment_slots=None):
        """The reference's segment store grows on demand (DynamicArray2D.add_rows / add_cols, utils.py:113-135); the
        device's pool has a fixed capacity.  This re-creates the engine with a larger pool (default: twice the segments)
        and hands the state over -- a host round trip.  Pools left at their default size grow by themselves (see
        _grow_if_needed); an explicit `segment_capacity=` is a hard limit and overflowing it raises CapacityError."""
        if self._fused:
            raise RuntimeError("this TemporalMemory is fused into a HierarchicalTemporalMemory; call its grow_pool()")
        eng = self._ensure_engine(1)
        retire_states(eng)
        dp = self.distal_projection
        dp.segment_capacity = int(segment_capacity or 2 * eng.segment_capacity)
        dp.segment_slots = int(segment_slots or eng.segment_slots)
        bigger = Engine(0, self.column_dim, self.cell_dim, eng.active_columns, distal=dp, seed=self.seed, device=self.device)
        if eng.steps:
            bigger.import_tm_state(eng.export_tm_state())
        bigger._auto_grow = getattr(eng, "_auto_grow", False)
        self._bind(bigger, False)
        self._last_ref = None

    def get_empty_state(self):
        """networks.py:59-65."""
        st = TemporalMemory.State.__new__(TemporalMemory.State)
        object.__setattr__(st, "_engine", None)
        object.__setattr__(st, "_step", -1)
        object.__setattr__(st, "_cache", dict(
            active_cell=(np.empty(0, dtype=np.int32), np.empty(0, dtype=np.int32)), winner_cell=None,
            cell_activation=np.zeros((self.column_dim, self.cell_dim), dtype=np.bool_),
            cell_prediction=np.zeros((self.column_dim, self.cell_dim), dtype=np.bool_),
            active_column_bursting=np.empty(0, dtype=np.bool_), distal_state=None))
        return st

    def flatten_cell(self, cell):
        """networks.py:67-71."""
        if cell is None:
            return None
        assert len(cell) == 2 and len(cell[0].shape) == 1
        return cell[0] * self.cell_dim + cell[1]

    @property
    def last_state(self):
        """networks.py:57,127.  Held weakly: the State of the latest step is only read back from
        the device if somebody asks for it."""
        if not self._own_distal:
            return self._host_last if self._host_last is not None else self._empty_state
        st = self._last_ref() if self._last_ref is not None else None
        if st is None:
            if self._engine is None or self._engine.steps == 0:
                return self._empty_state
            st = self._new_state(self._last_cols)
        return st

    def _new_state(self, active_column=None):
        st = self.State(self._engine, self._engine.steps, active_column)
        self._last_ref, self._last_cols = weakref.ref(st), active_column
        return st

    def process(self, sp_state, prev_state=None, learning=True, return_winner_cell=True, epsilon=1e-8):
        """networks.py:91-128.  `prev_state`: None / this object's `last_state` (the previous step's state lives in
        device memory), or any State this object returned earlier -- its fields are then written back as the device's
        previous step (a host round trip).  `epsilon` (the tolerance of the best-matching / least-used ties, as
        float32): any value in (0, 1]."""
        if not self._own_distal:
            return self._process_host(sp_state, prev_state, learning, return_winner_cell, epsilon)
        adopt = None
        if prev_state is not None and prev_state is not self.last_state:
            d = prev_state.distal_state                                     # (reading the fields materialises a lazy State)
            adopt = (prev_state.cell_prediction, prev_state.cell_activation,
                     None if prev_state.winner_cell is None else self.flatten_cell(prev_state.winner_cell), d)
        if not 0.0 < epsilon <= 1.0:
            raise NotImplementedError("epsilon must lie in (0, 1]")
        if self._fused:
            raise RuntimeError("this TemporalMemory is fused into a HierarchicalTemporalMemory; call its process()")
        active_column = np.asarray(sp_state.active_column, dtype=np.int64)
        eng = self._ensure_engine(max(len(active_column), 1))
        retire_states(eng)
        if len(active_column) > eng.active_columns:          # the reference takes any number of columns: a larger engine,
            bigger = Engine(0, self.column_dim, self.cell_dim, max(len(active_column), 2 * eng.active_columns),   # same state
                            distal=self.distal_projection, seed=self.seed, device=self.device)
            if eng.steps:
                bigger.import_tm_state(eng.export_tm_state())
            self._bind(bigger, False)
            eng = bigger
        if getattr(eng, "_epsilon", 1e-8) != epsilon:
            eng.set_epsilon(epsilon)
            eng._epsilon = epsilon
        if _grow_if_needed(eng, max(len(active_column), 1)):
            self.grow_pool(*eng._grow_to)
            eng = self._engine
            if getattr(eng, "_epsilon", 1e-8) != epsilon:
                eng.set_epsilon(epsilon)
                eng._epsilon = epsilon
        if adopt is not None:
            eng.import_prev_state(*adopt)
        eng.tm_step(active_column, learning=learning, return_winner_cell=return_winner_cell)
        return self._new_state(active_column)

    def _process_host(self, sp_state, prev_state, learning, return_winner_cell, epsilon):
        """networks.py:91-128 on the host, for a `distal_projection=` object that lives there: its `process` / `update` /
        `get_jittered_potential_info` are called exactly where the reference calls them.  The columns are processed in
        ascending order and the "least used" jitter is the keyed draw of the device (policies 1 and 3)."""
        from ._keyed import draw_unit, STREAM_LEAST_USED
        dp, C, K = self.distal_projection, self.column_dim, self.cell_dim
        eps = np.float32(epsilon)
        if prev_state is None:
            prev_state = self.last_state                                                          # :92-93
        caller_cols = np.asarray(sp_state.active_column, dtype=np.int64)
        order = np.argsort(caller_cols, kind="stable")
        active_column = caller_cols[order]
        pred = np.asarray(prev_state.cell_prediction)[active_column].reshape(len(active_column), K)    # :96
        bursting = ~pred.any(axis=1, keepdims=True)                                               # :97
        winner_cell = None
        if learning or return_winner_cell:
            if prev_state.distal_state is None:                                                   # :74-75
                column_matching = np.zeros((len(active_column), 1), dtype=np.bool_)
                best = np.zeros((len(active_column), K), dtype=np.bool_)
            else:                                                                                 # :76-82
                cell_max, _ = dp.get_jittered_potential_info(prev_state.distal_state)
                cell_max = np.asarray(cell_max, dtype=np.float32).reshape(C, K)[active_column]
                column_max = cell_max.max(axis=1, keepdims=True) if len(active_column) else cell_max[:, :1]
                column_matching = column_max >= dp.segment_matching_threshold
                best = np.abs(cell_max - column_max) < eps
            count = np.asarray(dp.bundle_segments).reshape(C, K)[active_column].astype(np.float32)    # :85-86
            flat = active_column[:, None] * K + np.arange(K)
            jit = (count.astype(np.float64) + draw_unit(self.seed, STREAM_LEAST_USED, self._host_step, flat)).astype(np.float32)   # :87
            least = np.abs(jit - jit.min(axis=1, keepdims=True)) < eps if len(active_column) else jit.astype(np.bool_)   # :88
            winner = pred | (bursting & np.where(column_matching, best, least))                   # :102
            rows, cells = np.where(winner)                                                        # :103-104
            winner_cell = (active_column[rows], cells)
        if learning:                                                                              # :106-113
            column_punishment = np.ones(C, dtype=np.bool_)
            column_punishment[active_column] = False
            dp.update(prev_state.distal_state, np.asarray(prev_state.cell_activation).reshape(-1), self.flatten_cell(winner_cell),
                      np.repeat(column_punishment, K), winner_input=self.flatten_cell(prev_state.winner_cell), epsilon=epsilon)
        activated = pred | bursting                                                               # :115
        rows, cells = np.where(activated)
        active_cell = (active_column[rows], cells)
        cell_activation = np.zeros((C, K), dtype=np.bool_)
        cell_activation[active_column] = activated                                                # :118-119
        distal_state = dp.process(self.flatten_cell(active_cell), return_jittered_potential_info=return_winner_cell)   # :121
        cell_prediction = np.asarray(distal_state.prediction).reshape(C, K) > epsilon             # :122
        caller_bursting = np.empty_like(bursting)
        caller_bursting[order] = bursting                                                         # (per column, in the caller's order)
        st = TemporalMemory.State.__new__(TemporalMemory.State)
        object.__setattr__(st, "_engine", None)
        object.__setattr__(st, "_step", -1)
        object.__setattr__(st, "_cache", dict(active_cell=active_cell, winner_cell=winner_cell, cell_activation=cell_activation,
                                              cell_prediction=cell_prediction, active_column_bursting=caller_bursting,
                                              distal_state=distal_state))
        self._host_last = st
        self._host_step += 1
        return st

    compute = process


class _DistalState:
    """PredictiveProjection.State (projections.py:195-203)."""

    def __init__(self, d):
        self.__dict__.update(d)


class HierarchicalTemporalMemory:
    """networks.py:131-149.  With default (or bithtm_amd) components both layers share ONE device
    engine and a timestep is a single C-ABI call (`htm_step`)."""

    def __init__(self, input_dim, column_dim, cell_dim, active_columns=None, spatial_pooler=None,
                 temporal_memory=None, seed=0, device=0):
        if active_columns is None:
            active_columns = round(column_dim * 0.02)                                # networks.py:137
        self.column_dim = column_dim
        self.cell_dim = cell_dim
        self.active_columns = active_columns
        self.spatial_pooler = spatial_pooler or SpatialPooler(input_dim, column_dim, active_columns, device=device)    # :143
        self.temporal_memory = temporal_memory or TemporalMemory(column_dim, cell_dim, seed=seed, device=device)       # :144
        sp, tm = self.spatial_pooler, self.temporal_memory
        for name, obj in (("spatial_pooler", sp), ("temporal_memory", tm)):
            if not callable(getattr(obj, "process", None)):
                raise TypeError(f"{name} must have the reference's process method; got {type(obj).__name__}")
        # Both layers the device's own: ONE engine, one C call per timestep.  Any other object (example.py:7-12 swaps the
        # Temporal Memory this way) -- or a Temporal Memory whose distal projection lives on the host -- is called as
        # networks.py:146-149 calls it, each device-backed layer then stepping an engine of its own.
        self._engine = None
        if type(sp) is SpatialPooler and type(tm) is TemporalMemory and tm._own_distal:
            if sp._engine is not None or tm._engine is not None:
                raise ValueError("spatial_pooler / temporal_memory must not have been stepped on their own before fusing")
            proximal, boosting = sp._engine_parts()
            self._engine = Engine(sp.input_dim, column_dim, cell_dim, sp.active_columns,
                                  proximal=proximal, boosting=boosting, distal=tm.distal_projection,
                                  seed=tm.seed, device=device)
            sp._bind(self._engine, True)
            tm._bind(self._engine, True)

    @property
    def engine(self):
        return self._engine

    def grow_pool(self, segment_capacity=None, segment_slots=None):
        """A larger segment pool (default: twice the segments) under the same model: the engine is re-created and the
        whole state handed over (see TemporalMemory.grow_pool; utils.py:113-135 is what the reference does instead)."""
        eng = self._fused_engine("grow_pool()")
        state = self.state_dict()
        sp, tm = self.spatial_pooler, self.temporal_memory
        dp = tm.distal_projection
        dp.segment_capacity = int(segment_capacity or 2 * eng.segment_capacity)
        dp.segment_slots = int(segment_slots or eng.segment_slots)
        proximal, boosting = sp._engine_parts()
        if sp._own_proximal:
            proximal._engine, proximal._permanence = None, state["sp_permanence"]      # (the new engine uploads it)
        bigger = Engine(sp.input_dim, self.column_dim, self.cell_dim, sp.active_columns, proximal=proximal, boosting=boosting,
                        distal=dp, seed=tm.seed, device=tm.device)
        bigger._auto_grow = getattr(eng, "_auto_grow", False)
        self._engine = bigger
        sp._bind(bigger, True)
        tm._bind(bigger, True)
        self.load_state_dict(state)
        self._bank = None

    def process(self, input, learning=True):
        """networks.py:146-149."""
        eng = self._engine
        if eng is None:                             # a layer that is not the device's own: the reference's two calls
            sp_state = self.spatial_pooler.process(input, learning=learning)
            return sp_state, self.temporal_memory.process(sp_state, learning=learning)
        retire_states(eng)
        if _grow_if_needed(eng, self.active_columns):
            self.grow_pool(*eng._grow_to)
            eng = self._engine
        if not self.spatial_pooler._plain:          # plug-in objects on the host: SP phase by phase, then the TM with its winners
            sp_state = self.spatial_pooler._process_phases(input, learning, commit=False)
            eng.tm_step(sp_state.active_column, learning=learning)
            return sp_state, self.temporal_memory._new_state(sp_state.active_column)
        eng.step(input, learning=learning)
        sp_state = SpatialPooler.State(eng, eng.steps)
        tm_state = self.temporal_memory._new_state(None)
        return sp_state, tm_state

    compute = process

    # ---- checkpoint / resume (the reference has none).  The dictionary holds the
    # reference's own arrays: DenseProjection.permanence, ExponentialBoosting.duty_cycle, and the
    # SparseProjection / PredictiveProjection store + last State in the layout of oracle export_state.
    def _fused_engine(self, what):
        if self._engine is None:
            raise RuntimeError(f"{what} needs both layers on the device (one engine); this model has a layer or a distal projection that lives on the host")
        return self._engine

    def state_dict(self):
        eng = self._fused_engine("state_dict()")
        retire_states(eng)
        out = {"tm_" + k: np.asarray(v) for k, v in eng.export_tm_state().items()}
        out["sp_permanence"] = eng.get_permanence()
        out["sp_duty_cycle"] = eng.read_duty_cycle()
        return out

    def load_state_dict(self, state):
        eng = self._fused_engine("load_state_dict()")
        retire_states(eng)
        eng.set_permanence(np.asarray(state["sp_permanence"], dtype=np.float64))
        eng.write(L.F_DUTY_CYCLE, np.asarray(state["sp_duty_cycle"], dtype=np.float32), np.float32)
        eng.import_tm_state({k[3:]: v for k, v in state.items() if k.startswith("tm_")})
        self.temporal_memory._last_ref = None

    def save(self, path):
        np.savez_compressed(path, **self.state_dict())

    def load(self, path):
        with np.load(path) as z:
            self.load_state_dict({k: z[k] for k in z.files})

    def run(self, inputs, steps, learning=True, use_graph=True, pipeline=True, continuing=False):
        """`steps` timesteps over the rows of the boolean matrix `inputs`, cycled, with the input
        bank resident in device memory and no per-step host work (the loop of example.py:48-53).
        Returns nothing; read `temporal_memory.last_state` or call process() afterwards.  `continuing=True`: the
        caller streams its input in chunks and the next call is another run() on the same inputs (HTM_RUN_CONTINUE:
        the Spatial Pooler keeps working ahead across the calls; finish with a run() without it)."""
        eng = self._fused_engine("run()")
        if not self.spatial_pooler._plain:
            raise RuntimeError("run() keeps the whole loop on the device: not available with plug-in objects that live on the host")
        retire_states(eng)
        inputs = np.asarray(inputs, dtype=np.bool_)
        key = (inputs.shape, inputs.tobytes())
        bank = getattr(self, "_bank", None)
        if bank is None or bank[0] != key:
            self._bank = bank = (key, eng.upload_bank(inputs))
        # A pool left at its default size grows like the reference's arrays (utils.py:113-135): the run is cut into batches
        # the free segments are expected to last (2 x active_columns new segments per step: every column bursting, twice),
        # with a look at the pool between them.  An overflow inside a batch is still reported, never silent.
        auto, k = getattr(eng, "_auto_grow", False), self.active_columns
        done = 0
        while done < steps:
            n = steps - done
            if auto:
                if _grow_if_needed(eng, 2 * k, force_check=True):
                    if getattr(self, "_streaming", False):
                        raise RuntimeError("the segment pool has to grow in the middle of a streamed run(): end the stream (a run() without continuing=True) first")
                    self.grow_pool(*eng._grow_to)
                    eng = self._engine
                    self._bank = bank = (key, eng.upload_bank(inputs))
                    _grow_if_needed(eng, 2 * k, force_check=True)
                n = max(1, min(n, eng._free_segments // (2 * k) - 1))
            last = done + n >= steps
            eng.run(bank[1], inputs.shape[0], n, learning=learning, use_graph=use_graph, pipeline=pipeline, continuing=continuing and last)
            self._streaming = bool(continuing and last and pipeline)
            done += n
        self.temporal_memory._new_state(None)
        eng.check_capacity()


def _grow_if_needed(eng, per_step, every=128, force_check=False):
    """Pools whose size the user did not fix (`segment_capacity=None`) follow the reference's growing arrays
    (utils.py:113-135): every `every` host-fed steps the engine is asked how full it is (a synchronisation and a read-back:
    not more often), and True comes back -- with the sizes to grow to in eng._grow_to -- when the free segments would not
    last another `every` steps at `per_step` new segments each (every active column bursting), or a segment has filled three
    quarters of its slots.  (Never in the middle of a device-side batch.)"""
    if not getattr(eng, "_auto_grow", False) or not eng.has_tm:
        return False
    eng._since_check = getattr(eng, "_since_check", every) + 1
    if eng._since_check < every and not force_check:
        return False
    eng._since_check = 0
    info = eng.check_capacity()
    eng._free_segments = eng.segment_capacity - info.segments
    capacity = slots = None
    if eng._free_segments < (every + 1) * per_step:
        capacity = max(2 * eng.segment_capacity, info.segments + 4 * (every + 1) * per_step)
    if info.segments and eng.segment_slots < 512:
        nsyn = eng.read(L.F_SEG_NSYN, np.int32, info.local_segments)
        if int(nsyn.max(initial=0)) > 3 * eng.segment_slots // 4:      # (a segment gains at most one sample of synapses per step)
            slots = min(512, 2 * eng.segment_slots)
    eng._grow_to = (capacity, slots)
    return capacity is not None or slots is not None
