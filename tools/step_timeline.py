#!/usr/bin/env python3
"""Device-clock timeline of the pipelined schedule at the bench workload (configs[2]).

Every block of the four pipelined launches stamps the 100 MHz device wall clock when it starts and
when it ends (BITHTM_TRACE=1, include/bithtm_hip.h htm_trace_read).  This prints, for the last two
look-ahead steps of a graph-replayed run, each launch's span, the gap to the previous launch and
the time range of every role inside it -- the numbers DESIGN.md's schedule discussion quotes.

    python tools/step_timeline.py
"""
import argparse
import os
import sys

import numpy as np

os.environ["BITHTM_TRACE"] = "1"
WARMUP, STEPS = int(os.environ.get("TIMELINE_WARMUP", 1500)), int(os.environ.get("TIMELINE_STEPS", 35))      # (TIMELINE_WARMUP=0: the cold phase of a run from scratch)
os.environ["BITHTM_TRACE_UNTIL"] = str(WARMUP + STEPS - 2)      # the last two steps look ahead less: keep the steady state
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

LAUNCHES = ("tm_activate+sp_emit", "tm_mid+sp_learn", "tm_learn+sp_overlap", "tm_scan+sp_select")
TWO = os.environ.get("BITHTM_LEAN", "2") == "2"          # the default schedule: two launches per step
LEAN = ("tm_activate+tm_mid+sp_learn+sp_overlap" if TWO else "tm_activate+sp_learn", "tm_mid+sp_overlap", "tm_learn+tm_scan+sp_emit", "-")
LEAN_ON = os.environ.get("BITHTM_LEAN", "2") != "0"


def roles(launch, k, C):
    """(first block, last block + 1, name) of the roles of each launch, as bench's handle lays them out."""
    c256, cls = (C + 255) // 256, 96
    if LEAN_ON:                                      # the three-launch schedule (htm_pipeline.h); block counts as htm_create sets them
        n_act, n_learn = (k * 32 + 255) // 256, int(os.environ.get("BITHTM_LEAN_LEARN", 768 if os.environ.get("TIMELINE_WORKLOAD") == "large" else 512))
        n_ov = int(os.environ.get("BITHTM_LEAN_OVERLAP", 512))
        if launch == 0 and TWO:      # two launches per step: k_act_mid_rows, its roles in the grid's order
            m = int(os.environ.get("BITHTM_LEAN2_CLASSIFY", 192 if os.environ.get("TIMELINE_WORKLOAD") == "large" else 32))
            sizes = {"0": (("tm_activate", n_act),), "1": (("tm_mid block 0", 1), ("tm_mid classify", m)), "2": (("sp rows", k),), "3": (("sp_overlap", n_ov),)}
            out, at = [], 0
            for digit in os.environ.get("BITHTM_LEAN2_ORDER", "0312"):
                for name, n in sizes[digit]:
                    out.append((at, at + n, name))
                    at += n
            return tuple(out) + ((at, at + c256, "clear"), (at + c256, 4096, "zero match bits"))
        if launch == 0:
            return ((0, n_act, "tm_activate"), (n_act, n_act + k, "sp rows"), (n_act + k, n_act + k + c256, "sp duty"), (n_act + k + c256, 4096, "clear"))
        if launch == 1:
            return ((0, 1, "tm_mid block 0"), (1, 1 + 4 * cls, "tm_mid classify"), (1 + 4 * cls, 1 + 4 * cls + n_ov, "sp_overlap"), (1 + 4 * cls + n_ov, 4096, "zero match bits"))
        if launch == 2:
            return ((0, c256, "sp_emit"), (c256, c256 + n_learn, "tm_learn"), (c256 + n_learn, 4096, "tm_scan"))
        return ()
    if launch == 0:
        return ((0, c256, "sp_emit"), (c256, 4096, "tm_activate"))
    if launch == 1:
        cls4 = 4 * cls
        return ((0, 1, "tm_mid block 0"), (1, 1 + cls4, "tm_mid classify"), (1 + cls4, 1 + cls4 + k, "sp rows"), (1 + cls4 + k, 4096, "sp duty"))
    if launch == 2:
        return ((0, 256, "tm_learn"), (256, 4096, "sp_overlap"))
    return ((0, 64, "sp_select"), (64, 64 + c256, "clear"), (64 + c256, 4096, "tm_scan"))


def main():
    args = argparse.Namespace(steps=STEPS, warmup=WARMUP)
    # (TIMELINE_WORKLOAD=large: bench.py's large_pool leg -- 350 patterns; give TIMELINE_WARMUP=3500 for its learned state)
    w = dict(bench.LARGE_POOL if os.environ.get("TIMELINE_WORKLOAD") == "large" else bench.WORKLOAD)
    noisy, perm = bench.make_inputs(w)
    htm = bench.build_htm(w, perm, 0)
    eng = htm.engine
    bank = eng.upload_bank(noisy)
    eng.run(bank, noisy.shape[0], args.warmup, learning=True)
    eng.sync()                                       # (the pool's size has reached the host: the traced call picks the scan's form by it)
    print(f"segments: {eng.info().segments}; the traced call: {eng.run_plan(args.steps)}")
    # (TIMELINE_LEARNING=0: the traced steps run with learning off -- what the learning role's writes cost the others)
    eng.run(bank, noisy.shape[0], args.steps, learning=os.environ.get("TIMELINE_LEARNING", "1") != "0", use_graph=True, pipeline=True)
    eng.sync()
    t = eng.trace_read()
    k, C = htm.active_columns, w["column_dim"]
    events = []
    for slot in range(8):
        blocks = np.nonzero(t[slot][:, 0] > 0)[0]
        if len(blocks):
            events.append((t[slot][blocks, 0].min(), t[slot][blocks, 1].max(), slot, blocks))
    events.sort(key=lambda e: e[0])
    t0, prev = events[0][0], None
    for first, last, slot, blocks in events:
        gap = "" if prev is None or first - prev > 3000 else f"  gap {(first - prev) / 100:5.2f}"
        print(f"parity {slot // 4} {(LEAN if LEAN_ON else LAUNCHES)[slot % 4]:24s} {(first - t0) / 100:7.2f} .. {(last - t0) / 100:7.2f} us  span {(last - first) / 100:6.2f}{gap}")
        prev = last
        for lo, hi, name in roles(slot % 4, k, C):
            sel = blocks[(blocks >= lo) & (blocks < hi)]
            if len(sel):
                st, en = t[slot][sel, 0] - first, t[slot][sel, 1] - first
                dur = (en - st) / 100
                slow = sel[np.argsort(dur)[-3:]] - lo
                print(f"      {name:16s} {len(sel):5d} blocks  start {st.min() / 100:5.2f}..{st.max() / 100:5.2f}  end {en.min() / 100:5.2f}..{en.max() / 100:5.2f}"
                      f"  block time median {np.median(dur):5.2f} max {dur.max():5.2f} (slowest: {slow.tolist()})")
                if name == "tm_scan":
                    q = [np.median(dur[i::8]) for i in range(0)]
                    dec = [float(np.max(dur[i * len(sel) // 8:(i + 1) * len(sel) // 8])) for i in range(8)]
                    print("        max block time by eighth of the grid: " + " ".join(f"{x:5.2f}" for x in dec))


if __name__ == "__main__":
    main()
