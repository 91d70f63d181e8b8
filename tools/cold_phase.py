#!/usr/bin/env python3
"""Per-kernel device time (HIP events, one role per launch) over the first steps of a from-scratch run of
bench.py's workload: the cold phase in which every column bursts and ~k segments are created per step.

    python tools/cold_phase.py [--steps 250] [--window 25]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=250)
    ap.add_argument("--window", type=int, default=25)
    ap.add_argument("--pipeline", action="store_true")
    args = ap.parse_args()
    w = dict(bench.WORKLOAD)
    noisy, perm = bench.make_inputs(w)
    htm = bench.build_htm(w, perm, 0)
    eng = htm.engine
    bank = eng.upload_bank(noisy)
    eng.profile(True)
    names = None
    for t0 in range(0, args.steps, args.window):
        eng.run(bank, noisy.shape[0], args.window, learning=True, use_graph=False, pipeline=args.pipeline)
        prof = {n: 1e3 * ms / cnt for n, (ms, cnt) in eng.profile_read().items() if cnt}
        names = names or sorted(prof)
        info = eng.info()
        print(f"steps {t0:4d}-{t0 + args.window - 1:4d}  S={info.segments:6d} work={info.work_items:5d}  " +
              "  ".join(f"{n}={prof.get(n, 0):6.1f}" for n in names), flush=True)
    eng.profile(False)
    eng.check_capacity()


if __name__ == "__main__":
    main()
