#!/usr/bin/env python3
"""Where a permanence-row block's time goes in the first launch of the three-launch schedule (diagnostic build):

    BITHTM_EXTRA_FLAGS=-DBITHTM_ROWS_STAMPS python -m bithtm_amd.build --force && python tools/rows_phases.py

Thread 0 of the first 1 024 row blocks of k_act_rows stamps the device clock (100 MHz) at: 0 start, 1 step counter and
winner column here, 2 first pass's values here and updated, 3 first pass stored (issued) + mask words, 4 / 5 the same for
the second pass."""
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
    names = ["step + winner column", "pass 1 loaded", "pass 1 stored", "pass 2 loaded", "pass 2 stored"]
    for step in range(6):
        eng.run(bank, noisy.shape[0], 1, learning=True, use_graph=False, continuing=True)
        eng.sync()
        raw = eng.trace_read().reshape(-1)[7 * 8192: 7 * 8192 + 1024 * 8].reshape(1024, 8)
        t = raw[:, :6].astype(np.float64) / 100.0
        t0 = t[:, 0].min()
        ph = np.diff(t, axis=1)
        tot = t[:, 5] - t[:, 0]
        print(f"step {step}: blocks start {(t[:, 0] - t0).max():.2f} us apart, end {(t[:, 5] - t0).min():.2f}..{(t[:, 5] - t0).max():.2f} us (block time median {np.median(tot):.2f}, "
              f"90 % {np.percentile(tot, 90):.2f}); " + "  ".join(f"{n} {np.median(ph[:, i]):.2f}/{np.percentile(ph[:, i], 90):.2f}/{ph[:, i].max():.2f}" for i, n in enumerate(names)))


if __name__ == "__main__":
    main()
