#!/usr/bin/env python3
"""Where an emit block's time goes in the three-launch schedule (diagnostic build):

    BITHTM_EXTRA_FLAGS=-DBITHTM_EMIT_STAMPS python -m bithtm_amd.build --force && python tools/emit_phases.py

Every emit block of k_learn_scan_emit stamps the device clock (100 MHz) at: 0 start, 1 windowed histogram resolved,
2 own record published, 3 everybody's records read, 4 k-th key known, 5 winner list written; and leaves the number of
merged bucket entries.  Printed for several consecutive steps of bench.py's learned state."""
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
    names = ["resolve histogram", "own record", "read all records", "k-th key", "list + bitmap"]
    nb = (w["column_dim"] + 255) // 256
    for step in range(8):
        eng.run(bank, noisy.shape[0], 1, learning=True, use_graph=False, continuing=True)
        eng.sync()
        raw = eng.trace_read().reshape(-1)[: nb * 8].reshape(nb, 8)
        t = raw[:, :6].astype(np.float64) / 100.0
        ne, nraw = raw[:, 7] & 0xFFFFFFFF, raw[:, 7] >> 32
        ph = np.diff(t, axis=1)
        if int(ne.max()) > 160:
            ent = eng.trace_read().reshape(-1)[256 * 8: 256 * 8 + int(ne.max())]
            keys, cnt = ent & 0xFFFFFFFF, (ent >> 32) & 0xFFFF
            uk = np.unique(keys)
            print(f"    {len(uk)} distinct keys among {len(keys)} entries; (key bits above low_zero - smallest: weight, entries): "
                  + " ".join(f"{int(k - uk[0])}:{int(cnt[keys == k].sum())},{int((keys == k).sum())}" for k in uk[::-1]))
        t0 = t[:, 0].min()
        pub = t[:, 2] - t0                            # when each block's record was out
        print(f"    records published at median {np.median(pub):.2f} / last {pub.max():.2f} us (block {int(pub.argmax())}); all read {np.median(t[:, 3] - t0 - pub.max()):.2f} us after the last one; "
              f"k-th key known {np.median(t[:, 4] - t[:, 3]):.2f} us later")
        kth = (raw[:, 6].astype(np.float64) / 100.0 - t[:, 3])
        print(f"    (k-th key itself {np.median(kth):.2f}/{kth.max():.2f} us of that phase)")
        print(f"step {step}: merged entries {int(ne.max())}, raw keys per block max {int(nraw.max())} mean {nraw.mean():.1f}; blocks end {(t[:, 5] - t0).min():.2f}..{(t[:, 5] - t0).max():.2f} us; "
              + "  ".join(f"{n} {np.median(ph[:, i]):.2f}/{ph[:, i].max():.2f}" for i, n in enumerate(names)))


if __name__ == "__main__":
    main()
