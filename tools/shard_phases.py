#!/usr/bin/env python3
"""Where the two selects of the column-sharded step spend their time (diagnostic builds, all ranks in one process):

    BITHTM_EXTRA_FLAGS=-DBITHTM_EMIT_STAMPS python -m bithtm_amd.build --force && python tools/shard_phases.py emit
    BITHTM_EXTRA_FLAGS=-DBITHTM_SHARD_STAMPS python -m bithtm_amd.build --force && python tools/shard_phases.py select

emit: every block of rank 0's candidates kernel (k_sp_emit, EMIT_LOCAL) stamps the device clock at 0 start, 1 windowed
histogram resolved, 2 own record published, 3 everybody's records read, 4 k-th key known, 5 list written.
select: block 0 of rank 0's k_shard_select at 0 start, 1 keys loaded (the histogram zeroed meanwhile), 2 k-th key known
(8 / 9 / 10 inside: histogram ready, filled, bin picked), 3 own winners emitted, 5 end.  The model is brought to bench.py's learned state unsharded and handed over."""
import os
import sys

import numpy as np

os.environ["BITHTM_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "emit"
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    import bithtm_amd as B
    from bithtm_amd.distributed import LocalGroup
    w = dict(bench.WORKLOAD)
    noisy, perm = bench.make_inputs(w)
    C, I, K = w["column_dim"], w["input_dim"], w["cell_dim"]
    solo = bench.build_htm(w, perm, 0)
    solo.run(noisy, 1000)
    state = solo.state_dict()
    del solo
    group = LocalGroup(world, I, C, K, permanence=bench.LazyPermanence(C, I, 1),
                       make_parts=lambda r: dict(distal=B.PredictiveProjection(C * K, segment_slots=w["segment_slots"])))
    group.import_state({k[3:]: v for k, v in state.items() if k.startswith("tm_")}, state["sp_permanence"], state["sp_duty_cycle"])
    group.upload_bank(noisy)
    group.run(60)
    eng = group.engines[0]
    nb = (C // world + 255) // 256
    for step in range(8):
        group.run(1)
        eng.sync()
        raw = eng.trace_read().reshape(-1)
        if what == "emit":
            t = raw[: nb * 8].reshape(nb, 8)[:, :6].astype(np.float64) / 100.0
            ph = np.diff(t, axis=1)
            # (the usual path hands the whole threshold bin over; in brackets what the phase is when the bin is cut exactly)
            names = ["resolve histogram", "scan + count published [own record]", "cell words + the others' counts [read all records]",
                     "record written [k-th key]", "- [list + record]"]
            print(f"step {step}: blocks end {(t[:, 5] - t[:, 0].min()).min():.2f}..{(t[:, 5] - t[:, 0].min()).max():.2f} us; "
                  + "  ".join(f"{n} {np.median(ph[:, i]):.2f}/{ph[:, i].max():.2f}" for i, n in enumerate(names)))
        else:
            t = raw[[0, 1, 2, 3, 5]].astype(np.float64) / 100.0
            names = ["load keys + zero the histogram", "k-th key", "earlier ranks + own winners", "window for the next step"]
            sub = raw[8:11].astype(np.float64) / 100.0           # inside "k-th key": histogram zeroed, filled, bin picked
            print(f"step {step}: " + "  ".join(f"{n} {v:.2f}" for n, v in zip(names, np.diff(t))) + f"  total {t[4] - t[0]:.2f} us"
                  + f"   [k-th key: zero {sub[0] - t[1]:.2f}  fill {sub[1] - sub[0]:.2f}  pick {sub[2] - sub[1]:.2f}  inside the bin {t[2] - sub[2]:.2f}]")
    if what == "select":
        group.run(200)
        eng.sync()
        raw = eng.trace_read().reshape(-1)
        print(f"over ~270 steps, the long way was taken because: a rank had no hot list {int(raw[16])} (rank-steps), a hot list was over "
              f"the budget {int(raw[17])} (rank-steps), the lists held fewer than k keys {int(raw[18])} (steps; last sum {int(raw[19])}), the k-th hot key was below a list's floor {int(raw[20])} (steps)")
    info = eng.check_capacity()
    print(f"segments {info.segments}, rank 0 rows {info.local_segments}, select fallbacks {info.select_fallbacks} of {info.step_index} steps")


if __name__ == "__main__":
    main()
