#!/usr/bin/env python3
"""Where an overlap block's time goes in the middle launch of the three-launch schedule (diagnostic build):

    BITHTM_EXTRA_FLAGS=-DBITHTM_OVERLAP_STAMPS python -m bithtm_amd.build --force && python tools/overlap_phases.py

Every overlap block of k_mid_overlap stamps the device clock (100 MHz) at: 0 start, 1 LDS histogram zeroed (barrier),
2 mask rows and input here and counted, 3 keys stored and binned in LDS, 4 barrier, 5 histogram flushed (atomics issued)."""
import os
import sys

import numpy as np

os.environ["BITHTM_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    w = dict(bench.WORKLOAD)
    noisy, perm = bench.make_inputs(w)
    htm = bench.build_htm(w, perm, 0)
    eng = htm.engine
    bank = eng.upload_bank(noisy)
    eng.run(bank, noisy.shape[0], 1500, learning=True, continuing=True)
    names = ["zero LDS + barrier", "rows + input loaded, counted", "exp, keys stored, LDS atomics", "barrier", "flush"]
    nb = int(os.environ.get("BITHTM_LEAN_OVERLAP", 512))
    for step in range(6):
        eng.run(bank, noisy.shape[0], 1, learning=True, use_graph=False, continuing=True)
        eng.sync()
        raw = eng.trace_read().reshape(-1)[3 * 8192: 3 * 8192 + nb * 8].reshape(nb, 8)
        t = raw[:, :6].astype(np.float64) / 100.0
        t0 = t[:, 0].min()
        ph = np.diff(t, axis=1)
        print(f"step {step}: blocks start {(t[:, 0] - t0).max():.2f} us apart, end {(t[:, 5] - t0).min():.2f}..{(t[:, 5] - t0).max():.2f} us; "
              + "  ".join(f"{n} {np.median(ph[:, i]):.2f}/{ph[:, i].max():.2f}" for i, n in enumerate(names)))


if __name__ == "__main__":
    main()
