import os, sys, ctypes as C
import numpy as np
os.environ["BITHTM_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bithtm_amd import _lib
w = dict(bench.WORKLOAD)
noisy, perm = bench.make_inputs(w)
htm = bench.build_htm(w, perm, 0)
eng = htm.engine
bank = eng.upload_bank(noisy)
n = noisy.shape[0]
eng.run(bank, n, 1500, learning=True)
eng.sync()
lib = eng.lib
lib.htm_debug_trace.argtypes = [C.c_void_p, C.c_void_p]
def dump(tag):
    buf = np.zeros(1024, np.uint64)
    lib.htm_debug_trace(eng.h, buf.ctypes.data_as(C.c_void_p))
    t = buf.astype(np.int64)
    t0 = min(t[0:128:16].min(), t[256:272:2][t[256:272:2] > 0].min())
    print(tag)
    for b in range(8):
        r = t[b * 16: b * 16 + 8]
        print(f"  emit blk {b*32:4d}: " + " ".join(f"{(x - t0) / 100:6.2f}" for x in r), "us (start, resolved, record built, stored, polled, T known, scanned, end)")
    for g in range(8):
        print(f"  learn blk {g*64:4d}: start {(t[128 + 2*g] - t0) / 100:6.2f} end {(t[129 + 2*g] - t0) / 100:6.2f}")
    for g in range(8):
        print(f"  scan blk {g*256:4d}: start {(t[256 + 2*g] - t0) / 100:6.2f} end {(t[257 + 2*g] - t0) / 100:6.2f}")
for pipeline in (True, False):
    eng.run(bank, n, 21, learning=True, use_graph=False, pipeline=pipeline)
    eng.sync()
    dump(f"pipeline={pipeline}")
