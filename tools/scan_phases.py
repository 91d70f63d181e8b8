#!/usr/bin/env python3
"""Where a scan block's time goes (diagnostic build):

    BITHTM_EXTRA_FLAGS=-DBITHTM_SCAN_STAMPS python -m bithtm_amd.build --force && python tools/scan_phases.py

Every scan wave of the three-launch schedule's k_learn_scan_emit stamps, in its first iteration, the device clock
(100 MHz) at: 0 start, 1 bitmap staged,
2 synapse counts + first chunks here, 3 every synapse counted (cell words read), 4 matching segments published
(atomics, info words), 5 match bits stored.  Printed for the last step of bench.py's learned state: median wave and the
slowest waves, with what their segments were (matching segments, synapses on active columns)."""
import os
import sys

import numpy as np

os.environ["BITHTM_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from bithtm_amd import _lib as L  # noqa: E402


def main():
    w = dict(bench.WORKLOAD)
    noisy, perm = bench.make_inputs(w)
    htm = bench.build_htm(w, perm, 0)
    eng = htm.engine
    bank = eng.upload_bank(noisy)
    eng.run(bank, noisy.shape[0], 1500, learning=True)
    eng.run(bank, noisy.shape[0], int(os.environ.get("SCAN_PHASES_STEPS", 33)), learning=True, use_graph=True, pipeline=True)
    eng.sync()
    t = eng.trace_read().reshape(-1)[: 2048 * 4 * 8].reshape(2048 * 4, 8).astype(np.float64) / 100.0
    S = eng.info().segments
    nw = (S + 15) // 16
    # trace row = block * 4 + wave; the wave's group of 16 segments (role_scan, small pools): waves of a block are 256 blocks
    # apart.  The schedule's scan grid is smaller than the pool (every wave loops): the first iteration covers the first groups.
    n_scan = int(os.environ.get("BITHTM_LEAN_SCAN", 768))
    blk, wave = np.arange(n_scan * 4) // 4, np.arange(n_scan * 4) % 4
    grp = (blk // 256) * 1024 + wave * 256 + blk % 256
    nw = min(nw, n_scan * 4)
    rows = np.argsort(grp)[:nw]                                  # trace row of group 0, 1, ...
    t = t[rows]
    pot = eng.read(L.F_SEG_POTENTIAL, np.int32, S)
    pot = np.pad(pot, (0, max(0, nw * 16 - S)))[: nw * 16].reshape(nw, 16)
    matching = (pot >= 10).sum(1)
    t0 = t[:, 0].min()
    ph = np.diff(t[:, :6], axis=1)
    total = t[:, 5] - t[:, 0]
    names = ["stage bitmap + barrier", "counts + first chunks", "cell words, count", "publish matching", "match bits"]
    print(f"S={S} waves={nw}; launch-relative start {np.median(t[:, 0] - t0):.2f} (median), end max {(t[:, 5] - t0).max():.2f}")
    for sel, label in ((matching == 0, "waves without a matching segment"), (matching >= 8, "waves with 8+ matching segments")):
        if sel.sum() == 0:
            continue
        print(f"{label}: {int(sel.sum())}; wave time median {np.median(total[sel]):.2f} p95 {np.percentile(total[sel], 95):.2f} max {total[sel].max():.2f}")
        for i, nme in enumerate(names):
            print(f"    {nme:26s} median {np.median(ph[sel, i]):5.2f}  p95 {np.percentile(ph[sel, i], 95):5.2f}  max {ph[sel, i].max():5.2f}")
    # the whole wave (all of its iterations): slot 6 = device clock when it left, slot 7 = groups it took
    end, iters = t[:, 6] - t0, (t[:, 7] * 100.0).astype(int)
    print(f"waves by groups taken: " + ", ".join(f"{k}: {int((iters == k).sum())} (left at median {np.median(end[iters == k]):.2f}, max {end[iters == k].max():.2f})" for k in sorted(set(iters.tolist()))))
    last = np.argsort(end)[::-1][:12]
    print("last waves to leave: group block wave matching(first group)  first iteration  groups taken  left at")
    for wv in last:
        print(f"  {wv:5d} {int(rows[wv]) // 4:5d} {int(rows[wv]) % 4:2d} {int(matching[wv]):3d}  {total[wv]:6.2f}  {iters[wv]:2d}  {end[wv]:6.2f}")
    if os.environ.get("SCAN_PHASES_LAST") == "1":      # build with -DBITHTM_SCAN_STAMPS=2: slots 1..5 are of the last iteration
        two = iters >= 2
        ph2 = np.diff(t[:, 1:6], axis=1)
        print("last iteration of the waves that took two groups: " + "  ".join(f"{n} {np.median(ph2[two, i]):.2f}/{ph2[two, i].max():.2f}" for i, n in enumerate(names[1:])))
        for wv in last:
            print(f"  group {wv:5d}: began {t[wv, 1] - t0:6.2f}  " + "  ".join(f"{x:6.2f}" for x in ph2[wv]))
        return
    order = np.argsort(total)[::-1][:12]
    print("slowest waves: group block matching  start  " + "  ".join(n[:12] for n in names))
    for wv in order:
        print(f"  {wv:5d} {int(rows[wv]) // 4:5d} {int(matching[wv]):3d}  {t[wv, 0] - t0:5.2f}  " + "  ".join(f"{x:12.2f}" for x in ph[wv]))


if __name__ == "__main__":
    main()
