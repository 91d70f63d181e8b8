"""bench.py --gpus N (N > 1): the same workload column-sharded over N MI355X, one process per GPU
(launched by torch.distributed.run), one RCCL all-gather per timestep.  Strong scaling: the model
(65 536 columns x 32 cells) is fixed, each rank owns column_dim / N columns.

BITHTM_DIST_BACKEND=gloo and BITHTM_SINGLE_DEVICE=1 rehearse the multi-process flow on a box with
one GPU (records staged through host memory); the default is backend "nccl" (= RCCL) on
cuda:LOCAL_RANK with the records exchanged device-to-device."""

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def run_sharded(args):
    import torch
    import torch.distributed as dist
    import bench
    from bithtm_amd.distributed import ShardedHTM, env_rank_world
    rank, world, local_rank = env_rank_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    backend = os.environ.get("BITHTM_DIST_BACKEND", "nccl")
    device = 0 if os.environ.get("BITHTM_SINGLE_DEVICE") == "1" else local_rank
    torch.cuda.set_device(device)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend, rank=rank, world_size=world)

    w = dict(bench.WORKLOAD)
    if args.columns:
        w["column_dim"] = args.columns
    noisy, perm = bench.make_inputs(w)               # same seed on every rank: identical inputs and permanences
    C, I, K = w["column_dim"], w["input_dim"], w["cell_dim"]

    if backend == "nccl":
        gather = None                                # torch.distributed.all_gather_into_tensor on device buffers
    else:
        def gather(recv, send):                      # rehearsal path: stage through host memory
            host = send.cpu()
            parts = [torch.empty_like(host) for _ in range(world)]
            dist.all_gather(parts, host)
            recv.copy_(torch.cat(parts))
    import bithtm_amd as B
    htm = ShardedHTM(I, C, K, rank=rank, world=world, permanence=perm, seed=0, device=device, all_gather=gather,
                     distal=B.PredictiveProjection(C * K, segment_slots=w["segment_slots"]))
    del perm
    eng = htm.engine
    bank = eng.upload_bank(noisy)
    n_bank = noisy.shape[0]

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    htm.run(bank, n_bank, args.warmup)
    fence()
    t0 = time.perf_counter()
    htm.run(bank, n_bank, args.steps)
    fence()
    dt = time.perf_counter() - t0
    info = eng.check_capacity()
    worst = torch.tensor([dt], dtype=torch.float64)
    if backend == "nccl":
        worst = worst.cuda()
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    dt = float(worst.item())
    # every rank must have reached the same global state
    digest = torch.tensor([float(info.segments), float(info.winner_cells)], dtype=torch.float64)
    if backend == "nccl":
        digest = digest.cuda()
    lo, hi = digest.clone(), digest.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    consistent = bool((lo == hi).all().item())
    # per-kernel device time on this rank (HIP events on the engine's stream) over a short profiled replay;
    # every rank takes part, the exchange is collective
    roofline = None
    try:
        prof_steps = min(args.steps, 100)
        eng.profile(True)
        htm.run(bank, n_bank, prof_steps)
        prof = eng.profile_read()
        eng.profile(False)
        if rank == 0 and "tm_scan" in prof:
            store = eng.read_store()
            c0, c1 = htm.column_range
            own = (store["seg_cell"] // K >= c0) & (store["seg_cell"] // K < c1)
            nbytes = 4 * int(store["seg_nsyn"][own].sum()) + 8 * len(store["seg_nsyn"])
            ms, n = prof["tm_scan"]
            us = 1e3 * ms / max(n, 1)
            roofline = dict(bound="hbm", kernel="tm_scan (rank 0's own segments)", achieved=round(nbytes / us / 1e3, 1), peak=bench.HBM_PEAK_GBS,
                            unit="GB/s", frac=round(nbytes / us / 1e3 / bench.HBM_PEAK_GBS, 4), traffic=None, bytes_per_launch=nbytes,
                            avg_launch_us=round(us, 2),
                            kernel_us_per_step={k: round(1e3 * v[0] / prof_steps, 2) for k, v in prof.items()})
    except Exception as e:                           # the roofline object is a report, never a reason to lose the line
        bench.log(f"[bench_sharded] roofline pass skipped: {e}")
    out = None
    if rank == 0:
        steps_per_s = args.steps / dt
        out = dict(
            metric="HTM timesteps/sec (SP + TM, learning on), 65536 cols x 32 cells", value=round(steps_per_s, 1),
            unit="timesteps/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
            ms_per_step=round(1e3 * dt / args.steps, 5), higher_is_better=True, scaling="strong", vs_baseline=None,
            dtype="u32 bit-packed / f64 + f32 permanences", data="synthetic",
            config=dict(workload=f"configs[3]: 65536 columns x 32 cells sharded {world}-way, winner all-gather per step",
                        input_dim=I, column_dim=C, cell_dim=K, columns_per_gpu=C // world, active_columns=htm.active_columns,
                        patterns=w["patterns"], input_density=w["density"], flip_noise=w["noise"],
                        segments=int(info.segments), segment_slots=w["segment_slots"], backend=backend,
                        exchange_bytes_per_rank=int(eng.shard_record_bytes()), ranks_consistent=consistent),
            roofline=roofline, cpu_baseline=None)
    dist.barrier()
    dist.destroy_process_group()
    return out
