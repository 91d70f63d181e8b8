#!/usr/bin/env python3
"""Which scan blocks are slow, and what is in them?  Per 64-segment block of bench.py's learned state: device time in
the pipelined k_scan_sel launch (BITHTM_TRACE) beside the block's rows (two-chunk rows, synapses on active columns,
matching segments of the last step).  Diagnostic for the scan's tail."""
import os
import sys

import numpy as np

os.environ["BITHTM_TRACE"] = "1"
WARMUP, STEPS = 1500, 35
os.environ["BITHTM_TRACE_UNTIL"] = str(WARMUP + STEPS - 2)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    w = dict(bench.WORKLOAD)
    noisy, perm = bench.make_inputs(w)
    htm = bench.build_htm(w, perm, 0)
    eng = htm.engine
    bank = eng.upload_bank(noisy)
    eng.run(bank, noisy.shape[0], WARMUP, learning=True)
    eng.run(bank, noisy.shape[0], STEPS, learning=True, use_graph=True, pipeline=True)
    eng.sync()
    t = eng.trace_read()
    C = w["column_dim"]
    lo = 64 + (C + 255) // 256
    st = eng.read_store()
    n = np.asarray(st["seg_nsyn"])
    S = len(n)
    nb = (S + 63) // 64
    dur = np.zeros(nb)
    for slot in (3, 7):
        tt = t[slot][lo:lo + nb]
        dur += (tt[:, 1] - tt[:, 0]) / 100.0 / 2
    presyn = np.asarray(st["presyn"]).reshape(S, -1)
    info = eng.info()
    pad = nb * 64 - S
    two = np.pad((n > 32).astype(int), (0, pad)).reshape(nb, 64).sum(1)
    from bithtm_amd import _lib as L
    pot = np.pad(eng.read(L.F_SEG_POTENTIAL, np.int32, S), (0, pad)).reshape(nb, 64)
    actw = eng.read(L.F_CELL_ACTIVATION, np.uint32, C)
    col_on = actw != 0
    valid = np.arange(presyn.shape[1])[None, :] < n[:, None]
    colhits = np.pad((col_on[np.clip(presyn, 0, None) // w["cell_dim"]] & valid).sum(1), (0, pad)).reshape(nb, 64)
    feats = dict(two_chunk_rows=two, column_hits=colhits.sum(1), potential_sum=pot.sum(1), matching=(pot >= 10).sum(1), nsyn_sum=np.pad(n, (0, pad)).reshape(nb, 64).sum(1))
    order = np.argsort(dur)
    print(f"S={S} blocks={nb}  block time: median {np.median(dur):.2f} p90 {np.percentile(dur, 90):.2f} max {dur.max():.2f}")
    for name, f in feats.items():
        print(f"  corr(time, {name}) = {np.corrcoef(dur, f)[0, 1]:.3f}")
    print("slowest 24 blocks: id time " + " ".join(feats))
    for b in order[::-1][:24]:
        print(f"  {b:5d} {dur[b]:5.2f}  " + " ".join(f"{int(f[b]):6d}" for f in feats.values()))
    print("median 8 blocks:")
    for b in order[nb // 2 - 4: nb // 2 + 4]:
        print(f"  {b:5d} {dur[b]:5.2f}  " + " ".join(f"{int(f[b]):6d}" for f in feats.values()))
    slow = dur > np.percentile(dur, 97)
    print("blocks above p97:", np.nonzero(slow)[0].tolist())
    cells = np.pad(np.asarray(st["seg_cell"]), (0, pad), constant_values=-1).reshape(nb, 64)
    for b in order[::-1][:3]:
        print(f"  block {b}: owner cells {cells[b][:16].tolist()} nsyn {np.pad(n, (0, pad)).reshape(nb, 64)[b][:16].tolist()} pot {pot[b][:16].tolist()}")


if __name__ == "__main__":
    main()
