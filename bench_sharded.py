"""bench.py --gpus N (N > 1): the same workload column-sharded over N MI355X, one process per GPU
(launched by torch.distributed.run), one RCCL all-gather per timestep -- each rank's top-k candidate columns with
their cell words, 20 bytes each (27 KB per rank at 8-way) -- issued from inside the library on the engine's stream
(`htm_shard_step`: one C call per timestep).  Strong scaling: the model (65 536 columns x 32 cells) is fixed, each
rank owns column_dim / N columns, their cells and their cells' segments.

BITHTM_DIST_BACKEND=gloo and BITHTM_SINGLE_DEVICE=1 rehearse the multi-process flow on a box with one GPU (the
step is then split around a host-staged gather); the default is backend "nccl" (= RCCL over xGMI) on
cuda:LOCAL_RANK, torch.distributed being used only to hand the RCCL unique id to the ranks and for the barriers."""

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
        gather = None                                # ncclAllGather inside the library (htm_shard_step)
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
    # per-launch device time on this rank (HIP events on the engine's stream) over a short profiled replay; every
    # rank takes part, the exchange is collective.  The dominant launch is the arg-max over ALL launches of the step.
    roofline = None
    try:
        prof_steps = max(min(args.steps, 100), 20)
        eng.profile(True)
        htm.run(bank, n_bank, prof_steps)
        prof = {n: v for n, v in eng.profile_read().items() if v[1]}
        eng.profile(False)
        if rank == 0 and prof:
            store = eng.read_store()
            live = store["seg_gid"] >= 0
            cl = C // world
            k = htm.active_columns
            kl = min(k, cl)
            W = ((I + 127) // 128) * 4
            S_own, syn_own = int(live.sum()), int(store["seg_nsyn"][live].sum())
            info2 = eng.info()
            launch_bytes = {      # algorithmic bytes of this rank's share
                "shard_overlap": cl * W * 4 + W * 4 + cl * 4 + cl * 20 + 12 * C,
                "sp_select": cl * 8 + 4 * 4096 * 4,
                "shard_candidates": cl * 8 + 20 * kl + kl * (4 + 32 * 8),
                "shard_select": 5 * 8 * world * kl + 20 * k,
                "tm_mid": 13 * k + 8 * S_own + 12 * info2.matching_segments + (2 * 8 * I + W * 4) * k // world + 8 * cl,
                "tm_learn": int(16 * syn_own / max(S_own, 1) * info2.work_items) + 4 * S_own,
                "tm_scan": 4 * syn_own + 8 * S_own,
            }
            us = {n: 1e3 * ms / cnt for n, (ms, cnt) in prof.items()}
            dominant = max((n for n in us if n in launch_bytes), key=lambda n: us[n])
            ach = launch_bytes[dominant] / us[dominant] / 1e3
            roofline = dict(bound="hbm", kernel=f"{dominant} (rank 0's share)", achieved=round(ach, 1), peak=bench.HBM_PEAK_GBS,
                            unit="GB/s", frac=round(ach / bench.HBM_PEAK_GBS, 4), traffic=None,
                            bytes_per_launch=int(launch_bytes[dominant]), avg_launch_us=round(us[dominant], 2),
                            launches={n: dict(us=round(v, 2), bytes=int(launch_bytes.get(n, 0))) for n, v in us.items()})
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
            config=dict(workload=f"configs[3]: 65536 columns x 32 cells sharded {world}-way, winner-candidate all-gather per step",
                        input_dim=I, column_dim=C, cell_dim=K, columns_per_gpu=C // world, active_columns=htm.active_columns,
                        patterns=w["patterns"], input_density=w["density"], flip_noise=w["noise"],
                        segments=int(info.segments), segment_slots=w["segment_slots"], backend=backend,
                        exchange_bytes_per_rank=int(eng.shard_record_bytes()), exchange="in-library ncclAllGather" if backend == "nccl" else "host-staged gloo",
                        ranks_consistent=consistent),
            roofline=roofline, cpu_baseline=None)
    dist.barrier()
    dist.destroy_process_group()
    return out
