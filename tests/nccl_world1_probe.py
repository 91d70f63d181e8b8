"""Manual probe (not collected by pytest): backend "nccl" (= RCCL) with world_size 1 on the GPU box.
Checks that torch.distributed.all_gather_into_tensor accepts the exchange record buffers and that
engine kernels enqueued on torch's current stream interleave correctly with it."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29513")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    from bithtm_amd.distributed import ShardedHTM
    I, C, K = 256, 2048, 32
    np.random.seed(3)
    perm = np.random.randn(C, I) * 0.1
    # two shards of a 2-way model in this one process; shard 1's record is gathered through RCCL
    # (world 1 => the gather is a device copy done by RCCL), shard 0's is copied by torch
    members = [ShardedHTM(I, C, K, rank=r, world=2, permanence=perm, seed=3, all_gather=lambda a, b: None) for r in range(2)]
    n = members[0].send.numel()
    tmp = torch.empty(n, dtype=torch.uint8, device="cuda")
    rng = np.random.RandomState(4)
    for t in range(50):
        x = rng.rand(I) < 0.05
        for m in members:
            m.engine.shard_begin(m.send.data_ptr(), input_bits=x)
        dist.all_gather_into_tensor(tmp, members[1].send)          # RCCL on torch's stream
        for m in members:
            m.recv[:n].copy_(members[0].send)
            m.recv[n:].copy_(tmp)
            m.engine.shard_finish(m.recv.data_ptr())
    a, b = members[0].engine.check_capacity(), members[1].engine.check_capacity()
    assert a.segments == b.segments and a.winner_cells == b.winner_cells, (a.segments, b.segments)
    print(f"nccl world-1 probe OK: {a.segments} segments on both shards after 50 steps")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
