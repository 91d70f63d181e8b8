#!/usr/bin/env python3
"""All ranks of the column-sharded bench workload (configs[3]: 65 536 x 32) inside one process on one GPU
(bithtm_amd.distributed.LocalGroup: every sharded kernel, the all-gather as device copies): per-launch device time of
rank 0 -- what one rank's GPU would spend per timestep, the exchange itself excluded.

    python tools/shard_rehearsal.py [--world 8] [--steps 600]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--steps", type=int, default=0, help="untimed steps from scratch (default: 12 passes over the pattern bank)")
    ap.add_argument("--large", action="store_true", help="bench.py's large_pool workload (350 patterns: 1.6 M segments) instead of the headline's")
    args = ap.parse_args()
    import bithtm_amd as B
    from bithtm_amd.distributed import LocalGroup
    w = dict(bench.LARGE_POOL if args.large else bench.WORKLOAD)
    if not args.steps:
        args.steps = 12 * w["patterns"] if not args.large else 10 * w["patterns"]
    noisy, perm = bench.make_inputs(w)
    C, I, K = w["column_dim"], w["input_dim"], w["cell_dim"]
    group = LocalGroup(args.world, I, C, K, permanence=perm,
                       make_parts=lambda r: dict(distal=B.PredictiveProjection(C * K, segment_slots=w["segment_slots"], segment_capacity=w.get("segment_capacity"))))
    group.upload_bank(noisy)
    graph = "ROCP_TOOL_LIBRARIES" not in os.environ      # (rocprofv3 crashes inside hipGraph replay on this image)
    group.run(args.steps, use_graph=graph)
    eng = group.engines[0]
    eng.sync()
    t0 = time.perf_counter()
    group.run(200, use_graph=graph)
    eng.sync()
    wall = (time.perf_counter() - t0) / 200
    eng.profile(True)
    group.run(200)
    raw = {n: v for n, v in eng.profile_read().items() if v[1]}
    eng.profile(False)
    prof = {n: 1e3 * ms / cnt for n, (ms, cnt) in raw.items()}
    steady = {n: v for n, v in prof.items() if raw[n][1] >= 100}         # the launches every step has (a call's first and last step differ)
    info = eng.check_capacity()
    print(json.dumps(dict(workload="large_pool (350 patterns)" if args.large else "headline (50 patterns)", world=args.world, rank0_launch_us={n: round(v, 1) for n, v in prof.items()}, launches_per_step=len(steady),
                          rank0_kernels_us=round(sum(steady.values()), 1), all_ranks_wall_us_per_step=round(1e6 * wall, 1),
                          segments=info.segments, rank0_rows=info.local_segments, record_bytes=eng.shard_record_bytes(),
                          select_fallbacks=info.select_fallbacks, local_selects_cut_exactly=info.candidate_exact_steps, global_selects_among_hot_lists=info.hot_select_steps, steps=info.step_index)))


if __name__ == "__main__":
    main()
