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
    `distal_projection=` takes a bithtm_amd PredictiveProjection only (its state is the device's segment store).
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
        engine_states(engine).add(self)

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
    s = getattr(engine, "_live_states", None)
    if s is None:
        s = engine._live_states = weakref.WeakSet()
    return s


def retire_states(engine):
    """Called before a new step is enqueued: pin down every State somebody still holds."""
    for st in list(engine_states(engine)):
        st._materialize()
    engine_states(engine).clear()


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
            act = words_to_bool(eng.read(L.F_CELL_ACTIVATION, np.uint32, C), K)
            rows, cells = np.where(act[cols])                                        # networks.py:116-117
            bursting = np.empty(len(cols), dtype=np.bool_)      # (the device keeps it by ascending column)
            bursting[np.argsort(cols, kind="stable")] = eng.read(L.F_BURSTING, np.uint8, eng.active_columns)[:len(cols)].astype(np.bool_)
            out = dict(
                cell_activation=act,
                cell_prediction=words_to_bool(eng.read(L.F_CELL_PREDICTION, np.uint32, C), K),
                active_cell=(cols[rows], cells),
                active_column_bursting=bursting[:, None],
                winner_cell=None)
            if info.has_winner_cells:
                winner = words_to_bool(eng.read(L.F_WINNER_WORDS, np.uint32, C), K)
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
        self.distal_projection = _accept(distal_projection, PredictiveProjection, "distal_projection") \
            or PredictiveProjection(self.column_dim * self.cell_dim)                 # networks.py:55
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
        if adopt is not None:
            eng.import_prev_state(*adopt)
        eng.tm_step(active_column, learning=learning, return_winner_cell=return_winner_cell)
        return self._new_state(active_column)

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
        self.spatial_pooler = _accept(spatial_pooler, SpatialPooler, "spatial_pooler") \
            or SpatialPooler(input_dim, column_dim, active_columns, device=device)  # :143
        self.temporal_memory = _accept(temporal_memory, TemporalMemory, "temporal_memory") \
            or TemporalMemory(column_dim, cell_dim, seed=seed, device=device)       # :144
        sp, tm = self.spatial_pooler, self.temporal_memory
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

    def process(self, input, learning=True):
        """networks.py:146-149."""
        eng = self._engine
        retire_states(eng)
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
    def state_dict(self):
        eng = self._engine
        retire_states(eng)
        out = {"tm_" + k: np.asarray(v) for k, v in eng.export_tm_state().items()}
        out["sp_permanence"] = eng.get_permanence()
        out["sp_duty_cycle"] = eng.read_duty_cycle()
        return out

    def load_state_dict(self, state):
        eng = self._engine
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
        if not self.spatial_pooler._plain:
            raise RuntimeError("run() keeps the whole loop on the device: not available with plug-in objects that live on the host")
        eng = self._engine
        retire_states(eng)
        inputs = np.asarray(inputs, dtype=np.bool_)
        key = (inputs.shape, inputs.tobytes())
        bank = getattr(self, "_bank", None)
        if bank is None or bank[0] != key:
            self._bank = bank = (key, eng.upload_bank(inputs))
        eng.run(bank[1], inputs.shape[0], steps, learning=learning, use_graph=use_graph, pipeline=pipeline, continuing=continuing)
        self.temporal_memory._new_state(None)
        eng.check_capacity()


def _accept(obj, cls, name):
    if obj is None:
        return None
    if not isinstance(obj, cls):
        raise TypeError(f"{name} must be a bithtm_amd {cls.__name__} (host-side plug-ins are not run); "
                        f"got {type(obj).__name__}")
    return obj
