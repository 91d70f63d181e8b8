"""How the learned pool grows with the number of patterns (bench.py's workload otherwise): segments and synapses after every
pass over the pattern bank.  python tools/pool_growth.py [patterns] [passes] [capacity]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bench  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 350
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 4 << 20
w = dict(bench.WORKLOAD, patterns=P, noisy_copies=4, segment_capacity=cap)
noisy, perm = bench.make_inputs(w)
htm = bench.build_htm(w, perm, 0)
eng = htm.engine
bank = eng.upload_bank(noisy)
for a in range(passes):
    t0 = time.perf_counter()
    eng.run(bank, len(noisy), P, learning=True)
    eng.sync()
    dt = time.perf_counter() - t0
    info = eng.check_capacity()
    nsyn = eng.read(11, np.int32, info.segments).astype(np.int64)      # HTM_F_SEG_NSYN
    print(f"pass {a + 1}: S={info.segments} synapses={int(nsyn.sum())} dead={(nsyn < 15).sum()} matching={info.matching_segments} "
          f"work={info.work_items} {P / dt:.0f} steps/s plan={eng.run_plan(P)}", flush=True)
