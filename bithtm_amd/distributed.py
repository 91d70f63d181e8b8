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
    """R shards inside ONE process on ONE GPU, with the all-gather replaced by device copies:
    exercises every sharded kernel without multi-GPU hardware (tests/test_hip_sharded.py)."""

    def __init__(self, world, *args, **kw):
        self.members = [ShardedHTM(*args, rank=r, world=world, all_gather=lambda recv, send: None, **kw) for r in range(world)]

    def process(self, input_bits, learning=True):
        for m in self.members:
            m.engine.shard_begin(m.send.data_ptr(), input_bits=input_bits, learning=learning)
        n = self.members[0].send.numel()
        for m in self.members:
            for r, src in enumerate(self.members):
                m.recv[r * n:(r + 1) * n].copy_(src.send)
        for m in self.members:
            m.engine.shard_finish(m.recv.data_ptr(), learning=learning)
