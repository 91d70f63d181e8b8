#!/usr/bin/env python3
"""Row lengths of the segment store in bench.py's learned state: synapses per segment by 32-slot chunk, and how the
long rows sit together (per 64-segment scan block).  Diagnostic for the scan's tail."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    large = len(sys.argv) > 1 and sys.argv[1] == "large"       # (bench.py's large_pool leg: 350 patterns)
    w = dict(bench.LARGE_POOL if large else bench.WORKLOAD)
    noisy, perm = bench.make_inputs(w)
    htm = bench.build_htm(w, perm, 0)
    eng = htm.engine
    bank = eng.upload_bank(noisy)
    eng.run(bank, noisy.shape[0], 10 * w["patterns"] + (0 if large else 1500), learning=True)
    eng.sync()
    st = eng.read_store()
    n = np.asarray(st["seg_nsyn"])
    S = len(n)
    chunks = (n + 31) // 32
    print(f"S={S} mean synapses {n.mean():.1f}")
    thr = 15                                         # a row can match whatever its first chunk holds once its later synapses alone reach the threshold
    print(f"  rows of more than 32 synapses: {100.0 * (n > 32).mean():.1f} %; of more than {32 + thr} (their second chunk is always fetched): {100.0 * (n > 32 + thr).mean():.1f} %")
    print("  synapses per row, deciles: " + " ".join(str(int(x)) for x in np.percentile(n, np.arange(0, 101, 10))))
    # what the scan's first-chunk test lets through against the last step's active columns: a row goes on to its second chunk
    # if its synapses on active columns among the first 32 plus all its later synapses reach the matching threshold
    from bithtm_amd import _lib as L
    K, C = w["cell_dim"], w["column_dim"]
    active = np.zeros(C, bool)
    active[eng.read(L.F_ACTIVE_COLUMN, np.int32, htm.active_columns)] = True
    presyn = np.asarray(st["presyn"])[:, :32]
    hits = (active[np.clip(presyn, 0, None) // K] & (presyn >= 0)).sum(axis=1)
    can = hits + np.maximum(n - 32, 0) >= thr
    print(f"  first-chunk hits on active columns: mean {hits.mean():.2f}, deciles " + " ".join(str(int(x)) for x in np.percentile(hits, np.arange(0, 101, 10))))
    print(f"  rows that go on past the bitmap test (cell words looked up): {100.0 * can.mean():.1f} %; of them longer than 32 (second chunk fetched): {100.0 * (can & (n > 32)).mean():.1f} % of all rows")
    per_wave = (can & (n > 32))[: S // 16 * 16].reshape(-1, 16).any(axis=1)
    print(f"  16-row groups with at least one such row: {100.0 * per_wave.mean():.1f} %")
    for c in range(0, 6):
        print(f"  rows with {c} chunks: {int((chunks == c).sum())} ({100.0 * (chunks == c).mean():.2f} %)")
    blocks = chunks[: S // 64 * 64].reshape(-1, 64).max(axis=1)
    for c in range(1, 6):
        print(f"  64-segment blocks whose longest row has {c} chunks: {int((blocks == c).sum())}")
    waves = chunks[: S // 16 * 16].reshape(-1, 16).max(axis=1)
    for c in range(1, 6):
        print(f"  16-segment waves whose longest row has {c} chunks: {int((waves == c).sum())}")


if __name__ == "__main__":
    main()
