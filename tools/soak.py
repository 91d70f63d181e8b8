"""Soak at the bench workload: 300 000 timesteps of the batched run, twice from the same seed, with a capacity
check every 50 000 steps; prints the segment count, the number of select fallbacks and a digest of the final
segment store and duty cycles (the two runs must agree: the path is deterministic).

    python tools/soak.py            # the headline workload (50 patterns): twice as the library schedules it (two launches per step)
                                    # and once in the three-launch schedule -- three digests that must agree
    python tools/soak.py large      # bench.py's large_pool workload (350 patterns): 40 000 steps, twice as the library
                                    # schedules them and once each with fixed scan shares, in the three-launch and in the
                                    # four-launch schedule -- five digests that must all agree (different kernels, one result)
"""
import os, sys, time, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
large = len(sys.argv) > 1 and sys.argv[1] == "large"
w = dict(bench.LARGE_POOL, segment_capacity=3 << 20) if large else dict(bench.WORKLOAD)
chunks, per_chunk = (8, 5000) if large else (6, 50000)
variants = [{}, {}, {"BITHTM_SCAN_DYN": "0"}, {"BITHTM_LEAN": "1"}, {"BITHTM_LEAN": "0"}] if large else [{}, {}, {"BITHTM_LEAN": "1"}]
noisy, perm = bench.make_inputs(w)
digests = []
for rep, env in enumerate(variants):
    os.environ.update(env)
    htm = bench.build_htm(w, perm, 0)              # (the knobs are read when the handle is created)
    for key in env:
        os.environ.pop(key)
    eng = htm.engine
    bank = eng.upload_bank(noisy)
    t0 = time.perf_counter()
    for chunk in range(chunks):
        eng.run(bank, noisy.shape[0], per_chunk, learning=True, use_graph=True)
        info = eng.check_capacity()
        print(f"rep {rep} {env or ''} after {info.step_index} steps: S={info.segments} fallbacks={info.select_fallbacks} ({time.perf_counter() - t0:.1f}s) {eng.run_plan(per_chunk)}", flush=True)
    st = eng.read_store()
    hsh = hashlib.sha256()
    for key in ("seg_cell", "seg_nsyn", "presyn", "perm", "segcount"):
        hsh.update(np.ascontiguousarray(st[key]).tobytes())
    hsh.update(eng.read_duty_cycle().tobytes())
    digests.append(hsh.hexdigest())
    print("digest", digests[-1][:16], flush=True)
    del htm, eng
print("deterministic" if len(set(digests)) == 1 else "DIFFERENT")
