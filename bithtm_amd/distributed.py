"""Column-sharded HierarchicalTemporalMemory: one process per GPU, one all-gather per timestep.

Each rank owns a contiguous block of mini-columns (their Spatial Pooler rows, their cells and the
distal segments of those cells, in rows of its own) and runs its own engine handle; the only exchange is an
all-gather of one fixed-size record per rank and step -- the rank's candidate columns (a superset of its top-k)
with their cell words and a short list of its best ones, 30 bytes per slot (DESIGN.md "Multi-GPU"; the protocol is pinned by oracle/sharded.py and
tests/test_sharded_gloo.py).  torch is used for what it is here for: device
buffers, the current stream, and torch.distributed (backend "nccl" = RCCL over xGMI).
"""

import os

import numpy as np

from .engine import Engine
from .projections import DenseProjection, PredictiveProjection
from .regularizations import ExponentialBoosting


def shard_range(rank, world, column_dim):
    if column_dim % (64 * world):
        raise ValueError("column_dim must be a multiple of 64 * world_size")
    per = column_dim // world
    return rank * per, (rank + 1) * per


def env_rank_world():
    """RANK / WORLD_SIZE / LOCAL_RANK as torch.distributed.run exports them."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def merge_shard_states(parts, column_dim, cell_dim):
    """The ranks' `engine.export_tm_state()` dictionaries (local rows under global ids, own cells, own columns) put
    together into the state an unsharded handle exports -- what `import_tm_state` takes on any handle, sharded or not
    (a sharded one keeps the rows of its own cells).  Checkpoint / hand-off of a column-sharded run."""
    parts = sorted(parts, key=lambda p: int(p["column_range"][0]))
    S, E, K = int(parts[0]["S"]), int(parts[0]["slots"]), cell_dim
    out = dict(S=np.int64(S), slots=np.int64(E), step_index=parts[0]["step_index"],
               seg_cell=np.zeros(S, np.int32), seg_nsyn=np.zeros(S, np.int32), presyn=np.full((S, E), -1, np.int32),
               perm=np.full((S, E), -1.0, np.float32), segcount=np.zeros(column_dim * K, np.int32),
               prev_prediction=np.zeros((column_dim, K), np.bool_), prev_activation=parts[0]["prev_activation"],
               prev_winner=parts[0]["prev_winner"], has_prev_winner=parts[0]["has_prev_winner"], has_distal=parts[0]["has_distal"])
    has_distal = bool(parts[0]["has_distal"])
    if has_distal:
        out.update(segment_potential=np.zeros(S, np.int64), max_jittered_potential=np.zeros(column_dim * K, np.float32),
                   prediction=np.zeros(column_dim * K, np.float64))
    match = []
    seen = np.zeros(S, np.bool_)
    for p in parts:
        assert int(p["S"]) == S, "the ranks disagree on the number of segment ids"
        c0, c1 = (int(x) for x in p["column_range"])
        gid = np.asarray(p["seg_gid"])
        live = np.flatnonzero(gid >= 0)
        g = gid[live]
        assert not seen[g].any(), "a segment id is owned by two ranks"
        seen[g] = True
        out["seg_cell"][g], out["seg_nsyn"][g] = p["seg_cell"][live], p["seg_nsyn"][live]
        out["presyn"][g], out["perm"][g] = p["presyn"][live], p["perm"][live]
        out["segcount"][c0 * K:c1 * K] = p["segcount"][c0 * K:c1 * K]
        out["prev_prediction"][c0:c1] = p["prev_prediction"][c0:c1]
        if has_distal:
            out["segment_potential"][g] = p["segment_potential"][live]
            out["max_jittered_potential"][c0 * K:c1 * K] = p["max_jittered_potential"][c0 * K:c1 * K]
            out["prediction"][c0 * K:c1 * K] = p["prediction"][c0 * K:c1 * K]
            rows = np.asarray(p["matching_segment"])
            match.append((gid[rows], p["matching_segment_activation"], p["matching_segment_active"], p["matching_segment_jittered_potential"]))
    assert seen.all(), "a segment id is owned by no rank"
    if has_distal:
        seg = np.concatenate([m[0] for m in match]).astype(np.int64)
        order = np.argsort(seg, kind="stable")
        out.update(matching_segment=seg[order], matching_segment_activation=np.concatenate([m[1] for m in match])[order],
                   matching_segment_active=np.concatenate([m[2] for m in match])[order],
                   matching_segment_jittered_potential=np.concatenate([m[3] for m in match])[order])
    return out


class ShardedHTM:
    """The reference's HierarchicalTemporalMemory.process (networks.py:146-149) over `world` GPUs.

    `all_gather=None` (production): the exchange happens inside the library -- one C call per timestep
    (`htm_shard_step`), RCCL `ncclAllGather` on the engine's stream; the communicator is created here from a
    unique id that rank 0 hands to the others through torch.distributed (any backend).  `all_gather(recv, send)`
    given: the step is split around that callable (`htm_shard_begin` / `htm_shard_finish`): rehearsals over gloo,
    and several shards inside one process (LocalGroup)."""

    def __init__(self, input_dim, column_dim, cell_dim, rank, world, active_columns=None, permanence=None,
                 proximal=None, boosting=None, distal=None, seed=0, device=0, all_gather=None):
        import torch
        self.torch = torch
        if active_columns is None:
            active_columns = round(column_dim * 0.02)
        self.rank, self.world = rank, world
        self.column_dim, self.cell_dim, self.active_columns = column_dim, cell_dim, active_columns
        self.column_range = shard_range(rank, world, column_dim)
        if proximal is None:
            proximal = DenseProjection.__new__(DenseProjection)
            proximal.input_dim, proximal.output_dim = input_dim, column_dim
            proximal.permanence_threshold, proximal.permanence_increment, proximal.permanence_decrement = 0.0, 0.03, 0.015
            proximal._engine = None
            proximal._permanence = permanence          # full [C, I] matrix or any object sliceable by rows
        boosting = boosting or ExponentialBoosting(column_dim, active_columns)
        distal = distal or PredictiveProjection(column_dim * cell_dim)
        torch.cuda.set_device(device)
        # a stream of its own when the exchange is the library's (the default stream cannot be captured into a hipGraph); the
        # caller's current stream when the caller gathers (its collective is ordered on that stream)
        self.stream = torch.cuda.Stream(device=device) if all_gather is None else torch.cuda.current_stream()
        self.engine = Engine(input_dim, column_dim, cell_dim, active_columns, proximal=proximal, boosting=boosting,
                             distal=distal, seed=seed, device=device, stream=self.stream.cuda_stream,
                             shard_rank=rank, shard_world=world)
        self.all_gather = all_gather
        self.record_bytes = self.engine.shard_record_bytes()
        if all_gather is None:
            import torch.distributed as dist
            ids = [self.engine.shard_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            self.engine.shard_comm_init(ids[0])
        else:
            self.send = torch.zeros(self.record_bytes, dtype=torch.uint8, device=f"cuda:{device}")
            self.recv = torch.zeros(self.record_bytes * world, dtype=torch.uint8, device=f"cuda:{device}")

    def process(self, input_bits, learning=True):
        eng = self.engine
        if self.all_gather is None:
            eng.shard_step(input_bits=input_bits, learning=learning)
            return
        eng.shard_begin(self.send.data_ptr(), input_bits=input_bits, learning=learning)
        self.all_gather(self.recv, self.send)
        eng.shard_finish(self.recv.data_ptr(), learning=learning)

    compute = process

    # ---- checkpoint / hand-off: the state of the whole model, as an unsharded handle exports and imports it
    def export_tm_state(self):
        """Collective: every rank contributes its part (torch.distributed.all_gather_object; not a fast path)."""
        import torch.distributed as dist
        parts = [None] * self.world
        dist.all_gather_object(parts, self.engine.export_tm_state())
        return merge_shard_states(parts, self.column_dim, self.cell_dim)

    def import_state(self, tm_state, permanence, duty_cycle):
        """The whole model's state (HierarchicalTemporalMemory.state_dict of an unsharded run, or export_tm_state of a
        sharded one + the Spatial Pooler's arrays): this rank keeps its own columns' rows, cells and segments."""
        from . import _lib as L
        eng = self.engine
        c0, c1 = self.column_range
        eng.set_permanence(np.asarray(permanence)[c0:c1], row_begin=c0)
        eng.write(L.F_DUTY_CYCLE, np.asarray(duty_cycle, dtype=np.float32), np.float32)
        eng.import_tm_state(tm_state)

    def run(self, device_bank, n_inputs, steps, learning=True, use_graph=True, pipeline=True):
        eng = self.engine
        if self.all_gather is None:                 # the loop, the exchange and the graphs inside the library
            eng.shard_run(device_bank, n_inputs, steps, learning=learning, use_graph=use_graph, pipeline=pipeline)
            return
        send, recv = self.send.data_ptr(), self.recv.data_ptr()
        for _ in range(steps):
            eng.shard_begin(send, device_bank=device_bank, n_inputs=n_inputs, learning=learning)
            self.all_gather(self.recv, self.send)
            eng.shard_finish(recv, learning=learning)


class LocalGroup:
    """R shards inside ONE process on ONE GPU (no torch): every rank's engine enqueues on the device's default stream
    and the all-gather is R x R device copies inside the library (`htm_shard_group_step`).  Exercises every sharded
    kernel without multi-GPU hardware (tests/test_hip_sharded.py, bench.py's configs[4] leg)."""

    def __init__(self, world, input_dim, column_dim, cell_dim, active_columns=None, permanence=None, make_parts=None, seed=0,
                 device=0):
        import ctypes as C
        if active_columns is None:
            active_columns = round(column_dim * 0.02)
        self.world, self.column_dim, self.cell_dim, self.active_columns = world, column_dim, cell_dim, active_columns
        self.engines = []
        for r in range(world):
            parts = make_parts(r) if make_parts else {}
            proximal = parts.get("proximal")
            if proximal is None:
                proximal = DenseProjection.__new__(DenseProjection)
                proximal.input_dim, proximal.output_dim = input_dim, column_dim
                proximal.permanence_threshold, proximal.permanence_increment, proximal.permanence_decrement = 0.0, 0.03, 0.015
                proximal._engine, proximal._permanence = None, permanence
            # rank 0's engine creates a stream of its own, the others enqueue on it (one in-order stream for the whole group:
            # what a graph capture of the group's step needs -- the default stream cannot be captured)
            eng = Engine(input_dim, column_dim, cell_dim, active_columns, proximal=proximal,
                         boosting=parts.get("boosting") or ExponentialBoosting(column_dim, active_columns),
                         distal=parts.get("distal") or PredictiveProjection(column_dim * cell_dim),
                         seed=seed, device=device, stream=(self.engines[0].stream_handle() if r else None), shard_rank=r, shard_world=world)
            if r:
                eng._stream_owner = self.engines[0]
            self.engines.append(eng)
        self._handles = (C.c_void_p * world)(*[e.h for e in self.engines])
        self._banks = None
        self.lib = self.engines[0].lib

    @property
    def members(self):                      # (objects with .engine / .rank / .column_range, as ShardedHTM has them)
        from types import SimpleNamespace
        return [SimpleNamespace(engine=e, rank=r, column_range=e.column_range) for r, e in enumerate(self.engines)]

    def _check(self, rc):
        if rc < 0:
            errs = "; ".join(self.lib.htm_last_error(e.h).decode() for e in self.engines)
            raise RuntimeError(f"htm_shard_group_step failed ({rc}): {errs}")

    def process(self, input_bits, learning=True):
        import ctypes as C
        from .engine import pack_bits
        packed = pack_bits(input_bits, self.engines[0].words)
        self._check(self.lib.htm_shard_group_step(self._handles, self.world, None, 1, packed.ctypes.data_as(C.c_void_p), int(bool(learning))))
        for e in self.engines:
            e.steps += 1

    def export_tm_state(self):
        return merge_shard_states([e.export_tm_state() for e in self.engines], self.column_dim, self.cell_dim)

    def import_state(self, tm_state, permanence, duty_cycle):
        """See ShardedHTM.import_state: every rank is handed the whole model's state and keeps its own share."""
        from . import _lib as L
        for e in self.engines:
            c0, c1 = e.column_range
            e.set_permanence(np.asarray(permanence)[c0:c1], row_begin=c0)
            e.write(L.F_DUTY_CYCLE, np.asarray(duty_cycle, dtype=np.float32), np.float32)
            e.import_tm_state(tm_state)

    def upload_bank(self, inputs):
        import ctypes as C
        self._banks = (C.c_void_p * self.world)(*[e.upload_bank(inputs) for e in self.engines])
        self._n_inputs = len(inputs)

    def run(self, steps, learning=True, use_graph=True, pipeline=True, stepwise=False):
        """`steps` timesteps over the bank of upload_bank (every rank holds its copy): the loop inside the library
        (htm_shard_group_run: whole timesteps replayed as hipGraphs, each rank's next overlap computed beside its learning and
        scan).  stepwise: one htm_shard_group_step call per timestep instead."""
        if stepwise:
            for _ in range(steps):
                self._check(self.lib.htm_shard_group_step(self._handles, self.world, self._banks, self._n_inputs, None, int(bool(learning))))
        else:
            flags = (1 if use_graph else 0) | (0 if pipeline else 2)
            self._check(self.lib.htm_shard_group_run(self._handles, self.world, self._banks, self._n_inputs, int(steps), int(bool(learning)), flags))
        for e in self.engines:
            e.steps += steps
