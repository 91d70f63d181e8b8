#!/usr/bin/env python3
"""Headline benchmark: HTM timesteps/s (SP + TM, learning on) at 65 536 columns x 32 cells.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full timestep of the hot path (SpatialPooler.process + TemporalMemory.process
with learning, networks.py:146-149) over one synthetic input.  The input bank (Bernoulli-sparse
patterns, cycled, with flip noise: BASELINE.md section 4) is resident in HBM before the timed
region.  Prints ONE JSON line (rank 0).

Besides the contract fields the line carries
  roofline      the dominant kernel's achieved HBM GB/s: algorithmic bytes per launch (formulas in
                DESIGN.md) / its average launch time, measured here with HIP events on the
                engine's stream over a profiled replay of the same workload;
  cpu_baseline  the NumPy oracle (a port of the reference's CPU path) timed on this box's host
                cores from the same learned state, on a bounded sample of steps.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = dict(input_dim=1024, column_dim=65536, cell_dim=32, patterns=50, density=0.02, noise=0.005,
                noisy_copies=20, segment_slots=128)
PIPELINED_LAUNCHES = ("tm_activate+sp_emit", "tm_mid+sp_learn", "tm_learn+sp_overlap", "tm_scan+sp_select")
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_inputs(w, seed=0):
    """seed 0; pattern bank rand(P, I) < d; SP permanences randn(C, I) * 0.1
    from the same stream; step t feeds pattern (t mod P) XOR flip noise."""
    np.random.seed(seed)
    P, I, C = w["patterns"], w["input_dim"], w["column_dim"]
    bank = np.random.rand(P, I) < w["density"]
    perm = np.random.randn(C, I) * 0.1
    noisy = np.empty((P * w["noisy_copies"], I), dtype=np.bool_)
    for r in range(w["noisy_copies"]):
        for p in range(P):
            noisy[r * P + p] = bank[p] ^ (np.random.rand(I) < w["noise"])
    return noisy, perm


def build_htm(w, perm, device, column_range=None):
    import bithtm_amd as B
    C, I, K = w["column_dim"], w["input_dim"], w["cell_dim"]
    k = round(C * 0.02)
    proximal = B.DenseProjection.__new__(B.DenseProjection)        # skip the RNG draw: perm is given
    proximal.input_dim, proximal.output_dim = I, C
    proximal.permanence_threshold, proximal.permanence_increment, proximal.permanence_decrement = 0.0, 0.03, 0.015
    proximal._engine, proximal._permanence = None, perm
    sp = B.SpatialPooler(I, C, k, proximal_projection=proximal)
    tm = B.TemporalMemory(C, K, distal_projection=B.PredictiveProjection(C * K, segment_slots=w["segment_slots"]), seed=0,
                          device=device)
    return B.HierarchicalTemporalMemory(I, C, K, active_columns=k, spatial_pooler=sp, temporal_memory=tm, device=device)


def kernel_bytes(w, k, store):
    """Algorithmic HBM bytes per launch of each kernel ('Kernels and rooflines')."""
    C, I = w["column_dim"], w["input_dim"]
    W = ((I + 127) // 128) * 4
    nsyn = store["seg_nsyn"].astype(np.int64)
    S = len(nsyn)
    syn = int(nsyn.sum())
    return {
        # mask read + input + duty read + overlap/boosted/key writes
        "sp_overlap": C * W * 4 + W * 4 + C * 4 + C * (4 + 8 + 8),
        # k winner rows: float64 read + write, mask row rewrite
        "sp_learn": 2 * 8 * k * I + k * W * 4,
        # packed presynaptic ids of every segment + per-segment nsyn read and potential write
        "tm_scan": 4 * syn + 8 * S,
        "sp_select": 6 * C * 8, "sp_count": C * 8, "sp_emit": C * (8 + 4 + 4 + 12),
    }


def recorded_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r01_pmc_summary.json:
    rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of this same workload; KB units;
    FETCH_SIZE doubled as MI355X_MICROARCH.md section HBM prescribes for gfx950).  PMC counters cannot be
    read from inside this process, so this is a recorded value, not a live one."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    name = {"tm_scan": "k_tm_scan", "sp_overlap": "k_sp_overlap", "sp_learn": "k_tm_mid"}.get(kernel)
    try:
        d = json.load(open(path))
        key = next(k for k in d["FETCH_SIZE"] if name in k)          # template kernels: "void k_tm_scan<true, 6>"
        f = d["FETCH_SIZE"][key]["mean_last150_KB"]
        wr = d["WRITE_SIZE"][key]["mean_last150_KB"]
        return dict(traffic=int((2 * f + wr) * 1024), traffic_source="profiles/r01_pmc_summary.json (recorded PMC pass)")
    except Exception:
        return dict(traffic=None)


def cpu_baseline(w, htm, noisy, start_step, sample_steps):
    """Time the NumPy oracle on this host from the GPU's learned state (a port of the
    reference's CPU path: dense float64 `>=` + `&` + sum overlap, NumPy segment scan)."""
    from oracle import HTMOracle
    eng = htm.engine
    C, I, K = w["column_dim"], w["input_dim"], w["cell_dim"]
    ora = HTMOracle(I, C, K, seed=0, permanence=eng.get_permanence())
    ora.spatial_pooler.duty_cycle = eng.read_duty_cycle().copy()
    ora.temporal_memory.import_state(eng.export_tm_state())
    n_bank = noisy.shape[0]
    ora.step(noisy[start_step % n_bank])                 # untimed: page in / allocate
    t0 = time.perf_counter()
    for t in range(1, sample_steps + 1):
        ora.step(noisy[(start_step + t) % n_bank])
    dt = time.perf_counter() - t0
    return dict(value=sample_steps / dt, unit="timesteps/s", cores=1, kind="port",
                sample=f"{sample_steps} timesteps of the NumPy oracle from the GPU's learned state "
                       f"(S={ora.temporal_memory.S} segments), single-threaded NumPy")


def run_single(args):
    w = dict(WORKLOAD)
    if args.columns:
        w["column_dim"] = args.columns
    device = int(os.environ.get("LOCAL_RANK", "0"))
    t_setup = time.perf_counter()
    noisy, perm = make_inputs(w)
    htm = build_htm(w, perm, device)
    del perm
    eng = htm.engine
    k = htm.active_columns
    bank = eng.upload_bank(noisy)
    n_bank = noisy.shape[0]
    log(f"[bench] setup {time.perf_counter() - t_setup:.1f}s; warm-up {args.warmup} steps")
    # rocprofv3 crashes inside hipGraph replay on this image: under the profiler, launch eagerly
    under_profiler = "ROCP_TOOL_LIBRARIES" in os.environ
    if under_profiler and not args.no_graph:
        log("[bench] rocprofv3 detected: hipGraph replay switched off (eager launches of the same schedule)")
    use_graph = not args.no_graph and not under_profiler
    pipeline = not args.no_pipeline
    eng.run(bank, n_bank, args.warmup, learning=True, use_graph=use_graph, pipeline=pipeline)
    eng.sync()
    eng.check_capacity()
    t0 = time.perf_counter()
    eng.run(bank, n_bank, args.steps, learning=True, use_graph=use_graph, pipeline=pipeline)
    eng.sync()
    dt = time.perf_counter() - t0
    info = eng.check_capacity()
    steps_per_s = args.steps / dt
    log(f"[bench] {args.steps} steps in {dt:.3f}s = {steps_per_s:.0f} timesteps/s; S={info.segments}; "
        f"select fallbacks so far: {info.select_fallbacks} of {info.step_index} steps")

    # per-kernel device time (HIP events on the engine's stream), same workload, profiled replay
    prof_steps = min(args.steps, 300)
    eng.profile(True)
    eng.run(bank, n_bank, prof_steps, learning=True, use_graph=False, pipeline=False)     # one role per launch
    prof = eng.profile_read()
    eng.run(bank, n_bank, prof_steps, learning=True, use_graph=False, pipeline=True)      # the launches of the timed run
    prof_pipe = {n: v for n, v in eng.profile_read().items() if n in PIPELINED_LAUNCHES}
    eng.profile(False)
    store = eng.read_store()
    kb = kernel_bytes(w, k, store)
    per_step_us = {name: 1e3 * ms / prof_steps for name, (ms, n) in prof.items()}
    avg_us = {name: 1e3 * ms / max(n, 1) for name, (ms, n) in prof.items()}
    dominant = max((n for n in per_step_us if n in ("sp_overlap", "sp_learn", "tm_scan")), key=lambda n: per_step_us[n])
    achieved = kb[dominant] / (avg_us[dominant] * 1e-6) / 1e9
    per_step_us = {n: v for n, v in per_step_us.items() if n not in PIPELINED_LAUNCHES}
    log("[bench] per-step device time by kernel, one role per launch (us): " +
        ", ".join(f"{n}={v:.1f}" for n, v in sorted(per_step_us.items(), key=lambda kv: -kv[1])))
    pipe_us = {n: 1e3 * ms / max(cnt, 1) for n, (ms, cnt) in prof_pipe.items()}
    log("[bench] pipelined launches (us): " + ", ".join(f"{n}={pipe_us.get(n, 0):.1f}" for n in PIPELINED_LAUNCHES) +
        f"; sum {sum(pipe_us.values()):.1f} of {1e6 / steps_per_s:.1f} us per step")
    roofline = dict(bound="hbm", kernel=dominant, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 4), traffic=None,
                    bytes_per_launch=int(kb[dominant]), avg_launch_us=round(avg_us[dominant], 2),
                    whole_step_bytes=int(sum(kb[n] for n in ("sp_overlap", "sp_learn", "tm_scan", "sp_select", "sp_count", "sp_emit"))))
    roofline["whole_step_frac"] = round(roofline["whole_step_bytes"] * steps_per_s / 1e9 / HBM_PEAK_GBS, 4)
    if dominant == "tm_scan":
        # SURVEY.md section 8(d) prices the scan at 8 bytes per slot of every allocated segment (the reference's
        # unpacked store, 64 slots at this state); the packed store above needs 4 bytes per VALID synapse and
        # reads permanences only for matching segments.  Both are given; `achieved` is the smaller one.
        S = int(len(store["seg_nsyn"]))
        sb = 8 * S * 64
        roofline["survey_formula"] = dict(bytes_per_launch=sb, achieved=round(sb / (avg_us[dominant] * 1e-6) / 1e9, 1),
                                          frac=round(sb / (avg_us[dominant] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                          formula="8 * S * E with E = 64 (reference layout)")
    roofline.update(recorded_traffic(dominant))

    cpu = None
    if not args.no_cpu_baseline:
        cpu = cpu_baseline(w, htm, noisy, int(eng.info().step_index), args.cpu_steps)
        log(f"[bench] cpu baseline: {cpu['value']:.2f} timesteps/s")

    return dict(
        metric="HTM timesteps/sec (SP + TM, learning on), 65536 cols x 32 cells", value=round(steps_per_s, 1),
        unit="timesteps/s", n_gpus=1, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * dt / args.steps, 5),
        higher_is_better=True, scaling="strong", vs_baseline=None, dtype="u32 bit-packed / f64 + f32 permanences",
        data="synthetic",
        config=dict(workload="configs[2]: 65536 columns x 32 cells, SP + TM learning on, 1 MI355X",
                    input_dim=w["input_dim"], column_dim=w["column_dim"], cell_dim=w["cell_dim"], active_columns=k,
                    patterns=w["patterns"], input_density=w["density"], flip_noise=w["noise"],
                    segments=int(info.segments), segment_slots=w["segment_slots"], hip_graph=use_graph,
                    pipelined=pipeline),
        roofline=roofline, cpu_baseline=cpu,
        kernel_us_per_step={n: round(v, 2) for n, v in per_step_us.items()},
        pipelined_launch_us={n: round(v, 2) for n, v in pipe_us.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--columns", type=int, default=0, help="override column_dim (debugging)")
    ap.add_argument("--cpu-steps", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="one role per launch (what the profiled replay always does)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand without the launcher: start one process per GPU as a CHILD (this process has
        # not touched the GPU) and pass its exit code on
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from bench_sharded import run_sharded
        out = run_sharded(args)
    else:
        out = run_single(args)
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
