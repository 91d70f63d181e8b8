"""The host-fed step against the batched run at the bench workload (65 536 x 32): the same 20 000 timesteps once through
htm.run (two launches per step, input bank in HBM) and once through engine.step (one ctypes call per timestep, three launches,
the learning role and the scan held back into the next call) -- the digests of the final segment store and duty cycles must
agree, and be the same with BITHTM_STEP_SPLIT=0 (the host-fed step of rounds 3-4).

    python tools/soak_hostfed.py [steps]
"""
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from bithtm_amd.engine import pack_bits  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
w = dict(bench.WORKLOAD)
noisy, perm = bench.make_inputs(w)
packed = None
digests = []
for mode, env in (("run", {}), ("step", {}), ("step", {"BITHTM_STEP_SPLIT": "0"})):
    os.environ.update(env)
    htm = bench.build_htm(w, perm, 0)
    for key in env:
        os.environ.pop(key)
    eng = htm.engine
    if packed is None:
        packed = [pack_bits(x, eng.words) for x in noisy]
    t0 = time.perf_counter()
    if mode == "run":
        bank = eng.upload_bank(noisy)
        eng.run(bank, noisy.shape[0], steps, learning=True)
    else:
        for t in range(steps):
            eng.step(packed[t % len(packed)], learning=True)
    eng.sync()
    dt = time.perf_counter() - t0
    info = eng.check_capacity()
    st = eng.read_store()
    hsh = hashlib.sha256()
    for key in ("seg_cell", "seg_nsyn", "presyn", "perm", "segcount"):
        hsh.update(np.ascontiguousarray(st[key]).tobytes())
    hsh.update(eng.read_duty_cycle().tobytes())
    digests.append(hsh.hexdigest())
    print(f"{mode:4s} {env or ''}: {steps} steps in {dt:.2f} s ({steps / dt:.0f} timesteps/s), S={info.segments}, digest {digests[-1][:16]}", flush=True)
    del htm, eng
print("identical" if len(set(digests)) == 1 else "DIFFERENT")
