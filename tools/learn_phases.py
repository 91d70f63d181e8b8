#!/usr/bin/env python3
"""Cycles per phase of the learning role (diagnostic build: BITHTM_EXTRA_FLAGS=-DBITHTM_LEARN_STAMPS python -m
bithtm_amd.build --force) over windows of a from-scratch run of bench.py's workload.

    BITHTM_EXTRA_FLAGS=-DBITHTM_LEARN_STAMPS python -m bithtm_amd.build --force && python tools/learn_phases.py
"""
import os
import sys

import numpy as np

os.environ["BITHTM_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PHASES = ("fetch", "perm update", "pre-stage", "stage", "membership", "rank+write")


def main():
    w = dict(bench.WORKLOAD)
    noisy, perm = bench.make_inputs(w)
    htm = bench.build_htm(w, perm, 0)
    eng = htm.engine
    bank = eng.upload_bank(noisy)
    prev = np.zeros(16, np.int64)
    waves = 256 * 8                       # kLearnBlocks x RB / 64
    for t0 in range(0, 600, 50):
        eng.run(bank, noisy.shape[0], 50, learning=True, use_graph=False, pipeline=False)
        eng.sync()
        tr = eng.trace_read().reshape(-1)[:waves * 16].reshape(waves, 16).sum(axis=0)
        d = tr - prev
        prev = tr.copy()
        items, grow, iters, staged = d[8], d[9], d[10], d[11]
        cyc = d[:6]
        print(f"steps {t0:3d}-{t0 + 49:3d}: items/step={items / 50:7.1f} growing={grow / 50:7.1f} tries/grow={iters / max(grow, 1):.2f} "
              f"staged/try={staged / max(iters, 1):6.1f} | cycles per item: " +
              "  ".join(f"{n}={c / max(items, 1):8.0f}" for n, c in zip(PHASES, cyc)) + f"  total={cyc.sum() / max(items, 1):8.0f}", flush=True)


if __name__ == "__main__":
    main()
