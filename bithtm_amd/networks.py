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
  * plug-in arguments accept the parameter objects of bithtm_amd.projections /
    bithtm_amd.regularizations (same constructor signatures as the reference's); arbitrary
    user objects are rejected with TypeError instead of being called on the host.
"""

import weakref

import numpy as np

from . import _lib as L
from .engine import Engine, words_to_bool
from .projections import DenseProjection, PredictiveProjection
from .regularizations import ExponentialBoosting, GlobalInhibition


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

    def __init__(self, input_dim, column_dim, active_columns, proximal_projection=None, boosting=None,
                 inhibition=None, device=0):
        self.input_dim = input_dim
        self.column_dim = column_dim
        self.active_columns = active_columns
        self.device = device
        self.proximal_projection = _accept(proximal_projection, DenseProjection, "proximal_projection") \
            or DenseProjection(input_dim, column_dim)                                   # networks.py:22
        self.boosting = _accept(boosting, ExponentialBoosting, "boosting") \
            or ExponentialBoosting(column_dim, active_columns)                          # :23
        self.inhibition = _accept(inhibition, GlobalInhibition, "inhibition") or GlobalInhibition(active_columns)  # :24
        self._engine = None
        self._fused = False

    def _bind(self, engine, fused):
        self._engine, self._fused = engine, fused
        self.proximal_projection._engine = engine
        self.boosting._engine = engine
        self.proximal_projection._permanence = None      # now lives in device memory

    def _ensure_engine(self):
        if self._engine is None:
            self._bind(Engine(self.input_dim, self.column_dim, 0, self.active_columns,
                              proximal=self.proximal_projection, boosting=self.boosting, device=self.device), False)
        return self._engine

    def process(self, input, learning=True):
        """networks.py:26-35."""
        eng = self._ensure_engine()
        if self._fused:
            raise RuntimeError("this SpatialPooler is fused into a HierarchicalTemporalMemory; call its process()")
        retire_states(eng)
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
            cols = np.sort(np.asarray(cols, dtype=np.int64))
            act = words_to_bool(eng.read(L.F_CELL_ACTIVATION, np.uint32, C), K)
            rows, cells = np.where(act[cols])                                        # networks.py:116-117
            out = dict(
                cell_activation=act,
                cell_prediction=words_to_bool(eng.read(L.F_CELL_PREDICTION, np.uint32, C), K),
                active_cell=(cols[rows], cells),
                active_column_bursting=eng.read(L.F_BURSTING, np.uint8, eng.active_columns)[:len(cols)].astype(np.bool_)[:, None],
                winner_cell=None)
            if info.has_winner_cells:
                flat = eng.read(L.F_WINNER_CELL, np.int32, info.winner_cells).astype(np.int64)
                out["winner_cell"] = (flat // K, flat % K)                           # networks.py:103-104
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
        """networks.py:91-128.  `prev_state` may only be None or this object's `last_state`: the
        previous step's state lives in device memory."""
        if prev_state is not None and prev_state is not self.last_state:
            raise NotImplementedError("prev_state other than last_state is not supported")
        if epsilon != 1e-8:
            raise NotImplementedError("epsilon is fixed at 1e-8")
        if self._fused:
            raise RuntimeError("this TemporalMemory is fused into a HierarchicalTemporalMemory; call its process()")
        active_column = np.asarray(sp_state.active_column, dtype=np.int64)
        eng = self._ensure_engine(max(len(active_column), 1))
        retire_states(eng)
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
        self._engine = Engine(sp.input_dim, column_dim, cell_dim, sp.active_columns,
                              proximal=sp.proximal_projection, boosting=sp.boosting, distal=tm.distal_projection,
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

    def run(self, inputs, steps, learning=True, use_graph=True, pipeline=True):
        """`steps` timesteps over the rows of the boolean matrix `inputs`, cycled, with the input
        bank resident in device memory and no per-step host work (the loop of example.py:48-53).
        Returns nothing; read `temporal_memory.last_state` or call process() afterwards."""
        eng = self._engine
        retire_states(eng)
        inputs = np.asarray(inputs, dtype=np.bool_)
        key = (inputs.shape, inputs.tobytes())
        bank = getattr(self, "_bank", None)
        if bank is None or bank[0] != key:
            self._bank = bank = (key, eng.upload_bank(inputs))
        eng.run(bank[1], inputs.shape[0], steps, learning=learning, use_graph=use_graph, pipeline=pipeline)
        self.temporal_memory._new_state(None)
        eng.check_capacity()


def _accept(obj, cls, name):
    if obj is None:
        return None
    if not isinstance(obj, cls):
        raise TypeError(f"{name} must be a bithtm_amd {cls.__name__} (host-side plug-ins are not run); "
                        f"got {type(obj).__name__}")
    return obj
