"""bench.py --gpus N (N > 1): the same workload column-sharded over N MI355X, one process per GPU
(launched by torch.distributed.run), one RCCL all-gather per timestep -- each rank's candidate columns (a superset of its
top-k) with their cell words and the short list of its best ones, 30 bytes per slot (50 KB per rank at 8-way) -- issued from
inside the library on the engine's stream:
`htm_shard_run` runs the whole timed region as ONE C call, whole timesteps (the collective included) replayed as
hipGraphs.  Strong scaling: the model (65 536 columns x 32 cells) is fixed, each rank owns column_dim / N columns,
their cells and their cells' segments.

The protocol is bench.py's, so that the N = 1 and the N > 1 lines lie on one curve:
  * every rank brings the UNSHARDED model to the learned state in untimed setup (10 passes over the pattern bank on its own
    GPU: 15 ms, identical on every rank), and hands that state to its shard (state import into a column-sharded handle);
  * a parity leg: the sharded model and, on rank 0, the NumPy oracle step the same timesteps from that state -- the
    oracle's time is the line's cpu_baseline, its outputs the check of the sharded path at the full size;
  * R repetitions of [W warm-up steps, exactly K timed steps between two barrier + synchronize fences], max over the
    ranks per repetition, `value` = the median repetition.

BITHTM_DIST_BACKEND=gloo and BITHTM_SINGLE_DEVICE=1 rehearse the multi-process flow on a box with one GPU (the
step is then split around a host-staged gather); the default is backend "nccl" (= RCCL over xGMI) on
cuda:LOCAL_RANK, torch.distributed being used only to hand the RCCL unique id to the ranks and for the barriers."""

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def shard_launch_bytes(w, world, k, rows, syn, n_work, n_match):
    """Algorithmic HBM bytes per timestep of one rank's launches: cl own columns, `rows` own segments
    with `syn` synapses in all."""
    C, I = w["column_dim"], w["input_dim"]
    cl, kl, W = C // world, min(k, C // world), ((I + 127) // 128) * 4
    overlap = cl * W * 4 + W * 4 + cl * 4 + cl * 20
    blocks = (cl + 255) // 256
    hot = max(1, min(kl, 4096 // world // 2))       # entries of a rank's hot list (its floor is the bin of its hot-th key)
    return {
        "shard_overlap": overlap,
        # local select: the threshold bin from the run sums and one run's bins (two waves per block), own keys, the blocks'
        # counts; the candidates' cell words (32 x (maximum + count) each); the record (at least the kl candidates the rank
        # must offer, 20 B each, + its hot list, 10 B each); the zeroing of the step's dense words
        "shard_candidates": blocks * 2 * (16 * 64 + 4 * 64) * 4 + cl * 8 + 8 * blocks + kl * (4 + 32 * 8) + 20 * kl + 10 * hot + 12 * C,
        # the short way: every rank's hot keys, the own rank's hot slots and their words, the winners' dense words
        "shard_select": 8 * world * hot + (2 + 12) * hot + 20 * k,
        "tm_mid": 13 * k + 8 * rows + 12 * n_match + (2 * 8 * I + W * 4) * k // world + 8 * cl,
        "tm_learn+tm_scan+shard_overlap": int(16 * syn / max(rows, 1) * n_work) + 4 * rows + 4 * syn + 8 * rows + overlap,
        "tm_learn+tm_scan": int(16 * syn / max(rows, 1) * n_work) + 4 * rows + 4 * syn + 8 * rows,
        "tm_learn": int(16 * syn / max(rows, 1) * n_work) + 4 * rows,
        "tm_scan": 4 * syn + 8 * rows,
    }


def run_sharded(args):
    import torch
    import torch.distributed as dist
    import bench
    from bithtm_amd import _lib as L
    from bithtm_amd.distributed import ShardedHTM, env_rank_world
    from bithtm_amd.engine import bool_to_words
    rank, world, local_rank = env_rank_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    backend = os.environ.get("BITHTM_DIST_BACKEND", "nccl")
    device = 0 if os.environ.get("BITHTM_SINGLE_DEVICE") == "1" else local_rank
    torch.cuda.set_device(device)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend, rank=rank, world_size=world)
    log = bench.log if rank == 0 else (lambda *a: None)

    w = dict(bench.WORKLOAD)
    if args.columns:
        w["column_dim"] = args.columns
    noisy, perm = bench.make_inputs(w)               # same seed on every rank: identical inputs and permanences
    C, I, K = w["column_dim"], w["input_dim"], w["cell_dim"]
    n_bank = noisy.shape[0]
    t_setup = time.perf_counter()

    # ---- untimed setup: the learned state, reached unsharded on this rank's own GPU, then handed to the shard
    pretrain = args.pretrain if args.pretrain >= 0 else 10 * w["patterns"]
    solo = bench.build_htm(w, perm, device)
    del perm
    solo.engine.run(solo.engine.upload_bank(noisy), n_bank, pretrain, learning=True)
    state = solo.state_dict()
    del solo

    if backend == "nccl":
        gather = None                                # ncclAllGather inside the library (htm_shard_run)
    else:
        def gather(recv, send):                      # rehearsal path: stage through host memory
            host = send.cpu()
            parts = [torch.empty_like(host) for _ in range(world)]
            dist.all_gather(parts, host)
            recv.copy_(torch.cat(parts))
    import bithtm_amd as B
    htm = ShardedHTM(I, C, K, rank=rank, world=world, permanence=bench.LazyPermanence(C, I, 1), seed=0, device=device, all_gather=gather,
                     distal=B.PredictiveProjection(C * K, segment_slots=w["segment_slots"]))
    tm_state = {k[3:]: v for k, v in state.items() if k.startswith("tm_")}
    htm.import_state(tm_state, state["sp_permanence"], state["sp_duty_cycle"])
    eng = htm.engine
    k = htm.active_columns
    c0, c1 = htm.column_range
    bank = eng.upload_bank(noisy)
    ranks_seen = eng.shard_comm_size() if backend == "nccl" else dist.get_world_size()
    in_graph = backend == "nccl" and eng.shard_graph_ok()
    log(f"[bench_sharded] setup {time.perf_counter() - t_setup:.1f}s incl. {pretrain} untimed pre-training steps (unsharded, on every rank) and the "
        f"hand-over to {world} shards; S={eng.info().segments} segments, {eng.info().local_segments} rows on rank 0; communicator of {ranks_seen} ranks")

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    # ---- parity leg (and, on rank 0, the CPU baseline): n_check timesteps from the imported state, sharded on the GPUs and
    # as the NumPy oracle on rank 0's host
    n_check = max(1, min(args.cpu_steps, 40)) if not args.no_cpu_baseline else 0
    cpu = None
    if n_check:
        htm.run(bank, n_bank, n_check)
        fence()
        info = eng.check_capacity()
        mine = dict(active=eng.read(L.F_ACTIVE_COLUMN, np.int32, k), act=eng.read(L.F_CELL_ACTIVATION, np.uint32, C),
                    pred=eng.read(L.F_CELL_PREDICTION, np.uint32, C)[c0:c1], winner=eng.read(L.F_WINNER_CELL, np.int32, info.winner_cells),
                    S=info.segments, duty=eng.read_duty_cycle()[c0:c1])
        parts = [None] * world
        dist.all_gather_object(parts, mine)
        if rank == 0:
            from oracle import HTMOracle
            ora = HTMOracle(I, C, K, seed=0, permanence=state["sp_permanence"])
            ora.spatial_pooler.duty_cycle = np.array(state["sp_duty_cycle"], dtype=np.float32)
            ora.temporal_memory.import_state(tm_state)
            start = int(tm_state["step_index"])
            o_sp, o_tm = ora.step(noisy[start % n_bank])                 # untimed: page in / allocate
            t0 = time.perf_counter()
            for t in range(1, n_check):
                o_sp, o_tm = ora.step(noisy[(start + t) % n_bank])
            dt = time.perf_counter() - t0
            cpu = dict(value=(n_check - 1) / dt if n_check > 1 else None, unit="timesteps/s", cores=1, kind="port",
                       sample=f"{n_check - 1} timesteps of the NumPy oracle from the learned state (S={ora.temporal_memory.S} segments), single-threaded NumPy")
            try:
                per = C // world
                winners = o_tm.winner_cell[0] * K + o_tm.winner_cell[1]
                for r, p in enumerate(parts):
                    assert p["S"] == ora.temporal_memory.S, f"rank {r}: segment count"
                    assert np.array_equal(p["active"], o_sp.active_column), f"rank {r}: active columns"
                    assert np.array_equal(p["act"], bool_to_words(o_tm.cell_activation)), f"rank {r}: cell activation (replicated)"
                    assert np.array_equal(p["pred"], bool_to_words(o_tm.cell_prediction)[r * per:(r + 1) * per]), f"rank {r}: cell prediction (own columns)"
                    assert np.array_equal(p["winner"], winners), f"rank {r}: winner cells"
                    assert np.array_equal(p["duty"].view(np.int32), ora.spatial_pooler.duty_cycle[r * per:(r + 1) * per].view(np.int32)), f"rank {r}: duty cycle (own columns)"
                cpu.update(parity="ok", parity_checked=f"{n_check} sharded timesteps from the imported learned state ({'htm_shard_run' if backend == 'nccl' else 'htm_shard_begin / _finish around a gloo gather'}) against the oracle: segment count, "
                                                        f"active columns, cell activation, winner cells on every rank; predictions and duty cycle of every rank's own columns")
                log(f"[bench_sharded] parity: {n_check} sharded timesteps equal the oracle's on all {world} ranks; cpu baseline {cpu['value']:.2f} timesteps/s")
            except AssertionError as e:
                log(f"[bench_sharded] PARITY FAILURE against the oracle: {e}")
                cpu.update(parity=f"FAILED: {e}")
        fence()

    # ---- R repetitions of [W warm-up steps, exactly K timed steps]
    reps = args.reps if args.reps > 0 else max(3, min(15, -(-4000 // max(args.steps, 1))))
    times = []
    for _ in range(reps):
        htm.run(bank, n_bank, args.warmup)
        fence()
        t0 = time.perf_counter()
        htm.run(bank, n_bank, args.steps)
        fence()
        worst = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if backend == "nccl":
            worst = worst.cuda()
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        times.append(float(worst.item()))
    dt = float(np.median(times))
    info = eng.check_capacity()
    # every rank must have reached the same global state
    digest = torch.tensor([float(info.segments), float(info.winner_cells), float(info.step_index)], dtype=torch.float64)
    if backend == "nccl":
        digest = digest.cuda()
    lo, hi = digest.clone(), digest.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    consistent = bool((lo == hi).all().item())
    # per-launch device time on this rank (HIP events on the engine's stream) over a short profiled replay (eager; every
    # rank takes part, the exchange is collective).  The dominant launch is the arg-max over ALL launches of the step.
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
            info2 = eng.info()
            lb = shard_launch_bytes(w, world, k, int(live.sum()), int(store["seg_nsyn"][live].sum()), info2.work_items, info2.matching_segments)
            us = {n: 1e3 * ms / cnt for n, (ms, cnt) in prof.items()}
            steady = {n: v for n, v in us.items() if prof[n][1] >= prof_steps // 2 and n in lb}      # (the launches every step has)
            dominant = max(steady, key=lambda n: steady[n])
            ach = lb[dominant] / us[dominant] / 1e3
            kernel_of = {"shard_overlap": "k_shard_overlap", "shard_candidates": "k_sp_emit", "shard_select": "k_shard_select", "tm_mid": "k_mid_rows",
                         "tm_learn+tm_scan+shard_overlap": "k_learn_scan_overlap", "tm_learn+tm_scan": "k_learn_scan_emit", "tm_learn": "k_tm_learn", "tm_scan": "k_tm_scan"}
            roofline = dict(bound="hbm", kernel=f"{kernel_of.get(dominant, dominant)} ({dominant}, rank 0's share)", achieved=round(ach, 1), peak=bench.HBM_PEAK_GBS,
                            unit="GB/s", frac=round(ach / bench.HBM_PEAK_GBS, 4), traffic=None,
                            bytes_per_launch=int(lb[dominant]), avg_launch_us=round(us[dominant], 2),
                            rank0_kernels_us_per_step=round(sum(steady.values()), 1),
                            launches={n: dict(kernel=kernel_of.get(n, n), us=round(v, 2), bytes=int(lb.get(n, 0)), per_step=prof[n][1] >= prof_steps // 2) for n, v in us.items()})
            # (PMC counters of the sharded launches: recorded by tools/collect_profiles.sh from the two-rank rehearsal on one GPU)
            # (no recorded PMC pass of the sharded launches: rocprofv3 crashes on all ranks in one process, and a launcher that
            # starts the ranks is an exec chain the profiler's preload cannot follow on this pool -- traffic stays null)
            roofline.update(traffic=None, traffic_source="no PMC pass of the sharded launches could be recorded on a one-GPU box (profiles/README.md)")
    except Exception as e:                           # the roofline object is a report, never a reason to lose the line
        bench.log(f"[bench_sharded] roofline pass skipped: {e!r}")
    out = None
    if rank == 0:
        steps_per_s = args.steps / dt
        log(f"[bench_sharded] {reps} x {args.steps} timed steps on {world} ranks: median {steps_per_s:.0f} timesteps/s (max over ranks per repetition)")
        out = dict(
            metric="HTM timesteps/sec (SP + TM, learning on), 65536 cols x 32 cells", value=round(steps_per_s, 1),
            unit="timesteps/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
            ms_per_step=round(1e3 * dt / args.steps, 5), higher_is_better=True, scaling="strong", vs_baseline=None,
            dtype="u32 bit-packed / f64 + f32 permanences", data="synthetic",
            config=dict(workload=f"configs[3]: 65536 columns x 32 cells sharded {world}-way, winner-candidate all-gather per step",
                        input_dim=I, column_dim=C, cell_dim=K, columns_per_gpu=C // world, active_columns=k,
                        patterns=w["patterns"], input_density=w["density"], flip_noise=w["noise"], pretrain_steps=pretrain,
                        segments=int(info.segments), segment_slots=w["segment_slots"], backend=backend, ranks_seen=int(ranks_seen),
                        exchange_bytes_per_rank=int(eng.shard_record_bytes()),
                        exchange=("in-library ncclAllGather, " + ("captured in the step's hipGraph" if in_graph else "launched eagerly (the preflight found it not capturable)")) if backend == "nccl" else "host-staged gloo",
                        hip_graph=in_graph, launches_per_step=4, repetitions=reps, ranks_consistent=consistent),
            repetitions=[round(args.steps / t, 1) for t in times],
            roofline=roofline, cpu_baseline=cpu)
    dist.barrier()
    dist.destroy_process_group()
    return out
