"""Segment-scan stress (flavour of BASELINE.json configs[4]): a synthetic pre-populated pool --
every segment 32 valid synapses to uniformly random presynaptic cells, permanences U[0.3, 0.7] --
scanned with learning off, to show what `k_tm_scan` reaches when the pool is large enough to be
HBM-bound.  Prints one JSON line per pool size.

    python tools/scan_stress.py --segments 1000000 4000000 --slots 64
"""

import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--segments", type=int, nargs="+", default=[1000000])
    ap.add_argument("--columns", type=int, default=65536)
    ap.add_argument("--cells", type=int, default=32)
    ap.add_argument("--slots", type=int, default=64)
    ap.add_argument("--synapses", type=int, default=32)
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    import bithtm_amd as B
    from bithtm_amd.engine import Engine
    C, K, E, n_syn = args.columns, args.cells, args.slots, args.synapses
    N, k = C * K, round(C * 0.02)
    rng = np.random.RandomState(0)
    for S in args.segments:
        distal = B.PredictiveProjection(N, segment_capacity=S, segment_slots=E)
        eng = Engine(0, C, K, k, distal=distal, seed=0)
        presyn = np.full((S, E), -1, dtype=np.int32)
        perm = np.full((S, E), -1.0, dtype=np.float32)
        for s0 in range(0, S, 1 << 20):              # in slices: the float64 intermediates stay small
            s1 = min(S, s0 + (1 << 20))
            presyn[s0:s1, :n_syn] = rng.randint(0, N, size=(s1 - s0, n_syn), dtype=np.int32)
            perm[s0:s1, :n_syn] = rng.uniform(0.3, 0.7, size=(s1 - s0, n_syn)).astype(np.float32)
        seg_cell = rng.randint(0, N, size=S).astype(np.int32)
        st = dict(S=S, slots=E, step_index=0, seg_cell=seg_cell, seg_nsyn=np.full(S, n_syn, np.int32), presyn=presyn, perm=perm,
                  segcount=np.bincount(seg_cell, minlength=N).astype(np.int32),
                  prev_prediction=np.zeros((C, K), bool), prev_activation=np.zeros((C, K), bool),
                  prev_winner=np.zeros(0, np.int64), has_prev_winner=False, has_distal=False)
        eng.import_tm_state(st)
        del presyn, perm
        cols = np.sort(rng.choice(C, k, replace=False)).astype(np.int32)
        for _ in range(5):
            eng.tm_step(cols, learning=False)
        eng.sync()
        eng.profile(True)
        for _ in range(args.steps):
            eng.tm_step(cols, learning=False)
        prof = eng.profile_read()
        eng.profile(False)
        ms, n = next(prof[k] for k in ("tm_scan_wide", "tm_scan_large", "tm_scan") if k in prof)
        us = 1e3 * ms / n
        algo = 4 * S * n_syn + 8 * S
        info = eng.info()
        print(json.dumps(dict(columns=C, cells=K, segments=S, slots=E, synapses_per_segment=n_syn, scan_us=round(us, 2),
                              algorithmic_bytes=algo, achieved_GBps=round(algo / us / 1e3, 1),
                              frac_of_8TBps=round(algo / us / 1e3 / 8000.0, 4), matching_segments=info.matching_segments)),
              flush=True)
        del eng


if __name__ == "__main__":
    main()
