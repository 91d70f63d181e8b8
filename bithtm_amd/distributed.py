"""Column-sharded HierarchicalTemporalMemory: one process per GPU, one all-gather per timestep.

Each rank owns a contiguous block of mini-columns (their Spatial Pooler rows, their cells and the
distal segments of those cells, in rows of its own) and runs its own engine handle; the only exchange is an
all-gather of one fixed-size record per rank and step -- the rank's top-k candidate columns with their cell
words, 20 bytes each (DESIGN.md "Multi-GPU"; the protocol is pinned by oracle/sharded.py and
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
        self.stream = torch.cuda.current_stream()
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

    def run(self, device_bank, n_inputs, steps, learning=True):
        eng = self.engine
        if self.all_gather is None:
            for _ in range(steps):
                eng.shard_step(device_bank=device_bank, n_inputs=n_inputs, learning=learning)
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
            self.engines.append(Engine(input_dim, column_dim, cell_dim, active_columns, proximal=proximal,
                                       boosting=parts.get("boosting") or ExponentialBoosting(column_dim, active_columns),
                                       distal=parts.get("distal") or PredictiveProjection(column_dim * cell_dim),
                                       seed=seed, device=device, stream="default", shard_rank=r, shard_world=world))
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

    def upload_bank(self, inputs):
        import ctypes as C
        self._banks = (C.c_void_p * self.world)(*[e.upload_bank(inputs) for e in self.engines])
        self._n_inputs = len(inputs)

    def run(self, steps, learning=True):
        """`steps` timesteps over the bank of upload_bank (every rank holds its copy)."""
        for _ in range(steps):
            self._check(self.lib.htm_shard_group_step(self._handles, self.world, self._banks, self._n_inputs, None, int(bool(learning))))
        for e in self.engines:
            e.steps += steps
