"""Soak at the bench workload: 300 000 timesteps of the batched run, twice from the same seed, with a capacity
check every 50 000 steps; prints the segment count, the number of select fallbacks and a digest of the final
segment store and duty cycles (the two runs must agree: the path is deterministic).

    python tools/soak.py
"""
import os, sys, time, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
w = dict(bench.WORKLOAD)
noisy, perm = bench.make_inputs(w)
digests = []
for rep in range(2):
    htm = bench.build_htm(w, perm, 0)
    eng = htm.engine
    bank = eng.upload_bank(noisy)
    t0 = time.perf_counter()
    for chunk in range(6):
        eng.run(bank, noisy.shape[0], 50000, learning=True, use_graph=True)
        info = eng.check_capacity()
        print(f"rep {rep} after {info.step_index} steps: S={info.segments} fallbacks={info.select_fallbacks} ({time.perf_counter() - t0:.1f}s)", flush=True)
    st = eng.read_store()
    hsh = hashlib.sha256()
    for key in ("seg_cell", "seg_nsyn", "presyn", "perm", "segcount"):
        hsh.update(np.ascontiguousarray(st[key]).tobytes())
    hsh.update(eng.read_duty_cycle().tobytes())
    digests.append(hsh.hexdigest())
    print("digest", digests[-1][:16], flush=True)
    del htm, eng
print("deterministic" if digests[0] == digests[1] else "DIFFERENT")
