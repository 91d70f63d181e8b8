#!/usr/bin/env python3
"""Launch spans of many steady-state steps of the two-launch schedule (BITHTM_LEAN=1: the three-launch one; device clock, graph replay).

tools/step_timeline.py shows two consecutive steps in detail; the trace holds no more than two.  This repeats the run
with the traced pair moved along and prints, per step, the span of each launch and when the last block of each role
of the last launch ended -- which role the launch waited for, and how often; in the two-launch schedule the same for the
first launch (activation, overlap, middle role, winner rows).

    python tools/step_spans.py [pairs]
"""
import os
import sys

import numpy as np

os.environ["BITHTM_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

WARMUP = int(os.environ.get("BITHTM_SPANS_WARMUP", 1500))      # (the state: bench.py times at 8 604 steps by default, 979 at the driver's arguments)


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    w = dict(bench.WORKLOAD)
    noisy, perm = bench.make_inputs(w)
    c256, n_learn = (w["column_dim"] + 255) // 256, int(os.environ.get("BITHTM_LEAN_LEARN", 512))
    two = os.environ.get("BITHTM_LEAN", "2") == "2"
    k = round(w["column_dim"] * 0.02)
    n_act, n_ov, n_cls = (k * 32 + 255) // 256, int(os.environ.get("BITHTM_LEAN_OVERLAP", 512)), int(os.environ.get("BITHTM_LEAN2_CLASSIFY", 32))
    first_roles, at = [], 0                          # (k_act_mid_rows' roles in the grid's order)
    for digit in os.environ.get("BITHTM_LEAN2_ORDER", "0312"):
        n = {"0": n_act, "1": 1 + n_cls, "2": k, "3": n_ov}[digit]
        first_roles.append((at, at + n, {"0": "act", "1": "mid", "2": "rows", "3": "overlap"}[digit]))
        at += n
    rows = []
    for u in range(pairs):
        steps = 35 + 2 * u
        os.environ["BITHTM_TRACE_UNTIL"] = str(WARMUP + steps - 2)
        htm = bench.build_htm(w, perm, 0)
        eng = htm.engine
        bank = eng.upload_bank(noisy)
        eng.run(bank, noisy.shape[0], WARMUP, learning=True)
        eng.run(bank, noisy.shape[0], steps, learning=True, use_graph=True, pipeline=True)
        eng.sync()
        t = eng.trace_read()
        for parity in (0, 1):
            rec = {}
            for launch in range(3):
                tt = t[parity * 4 + launch]
                blocks = np.nonzero(tt[:, 0] > 0)[0]
                if not len(blocks):
                    continue
                first = tt[blocks, 0].min()
                rec[launch] = (first, tt[blocks, 1].max())
                roles = ((0, c256, "emit"), (c256, c256 + n_learn, "learn"), (c256 + n_learn, 4096, "scan")) if launch == 2 else first_roles if two and launch == 0 else ()
                for lo, hi, name in roles:
                    sel = blocks[(blocks >= lo) & (blocks < hi)]
                    rec[name] = (tt[sel, 1].max() - first) / 100 if len(sel) else 0.0
            launches = (0, 2) if two else (0, 1, 2)
            if all(i in rec for i in launches) and "scan" in rec:
                rows.append((rec[0][0], [(rec[i][1] - rec[i][0]) / 100 for i in launches], rec["emit"], rec["learn"], rec["scan"],
                             [rec.get(n, 0.0) for n in ("act", "overlap", "mid", "rows")]))
        del htm, eng
    if two:
        print("launch spans (us): act_mid_rows  learn_scan_emit | last block of act / overlap / mid / rows | of emit / learn / scan")
    else:
        print("launch spans (us): act_rows  mid_overlap  learn_scan_emit | last block of emit / learn / scan")
    for _, sp, e, l, s, fr in rows:
        who = max((e, "emit"), (l, "learn"), (s, "scan"))[1]
        if two:
            who0 = max(zip(fr, ("act", "overlap", "mid", "rows")))[1]
            print(f"  {sp[0]:6.2f} {sp[1]:6.2f} | " + " ".join(f"{x:6.2f}" for x in fr) + f"  <- {who0:7s} | {e:6.2f} {l:6.2f} {s:6.2f}  <- {who}")
        else:
            print(f"  {sp[0]:6.2f} {sp[1]:6.2f} {sp[2]:6.2f} | {e:6.2f} {l:6.2f} {s:6.2f}  <- {who}")
    a = np.array([r[1] for r in rows])
    print(f"mean spans over {len(rows)} steps: " + " ".join(f"{x:.2f}" for x in a.mean(axis=0)) + f" us; last launch: "
          f"median {np.median(a[:, -1]):.2f}, above 11 us in {int((a[:, -1] > 11).sum())} steps")


if __name__ == "__main__":
    main()
