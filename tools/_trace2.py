import os, sys, ctypes as C
import numpy as np
os.environ["BITHTM_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
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
eng.run(bank, n, 34, learning=True, use_graph=True, pipeline=True)
eng.sync()
buf = np.zeros(1024 + 8 * 4096 * 2, np.uint64)
lib.htm_debug_trace(eng.h, buf.ctypes.data_as(C.c_void_p))
t = buf[1024:].astype(np.int64).reshape(8, 4096, 2)
ev = []
for i in range(8):
    m = t[i][:, 0] > 0
    if m.any():
        ev.append((t[i][m, 0].min(), t[i][m, 1].max(), i))
ev.sort()
t0 = ev[0][0]
names = ["activate+overlap", "mid+select", "learn+emit", "scan+sp_learn"]
prev = None
for a, b, i in ev:
    print(f"parity {i // 4} {names[i % 4]:18s} start {(a - t0) / 100:7.2f} end {(b - t0) / 100:7.2f} span {(b - a) / 100:6.2f}" + (f" gap {(a - prev) / 100:6.2f}" if prev else ""))
    prev = b
e = buf[:128].astype(np.int64).reshape(8, 16)[:, :8]
cstart = [a for a, b, i in ev if i % 4 == 2]
print("emit phases of the last learn+emit launch (us after its first block start):", [f"{(x - max(cstart)) / 100:.2f}" for x in e[0]])
m = t[2 + 4 * (ev[-1][2] // 4)] if False else None
for i in (2, 6):
    mm = t[i][:, 0] > 0
    if mm.any():
        st, en = t[i][mm, 0], t[i][mm, 1]
        idx = np.nonzero(mm)[0]
        for lo, hi, nm in ((0, 256, "emit"), (256, 4096, "learn")):
            sel = (idx >= lo) & (idx < hi)
            print(f"  slot {i} {nm}: start {(st[sel].min() - st.min()) / 100:.2f}..{(st[sel].max() - st.min()) / 100:.2f} end {(en[sel].min() - st.min()) / 100:.2f}..{(en[sel].max() - st.min()) / 100:.2f}")
            if nm == "learn":
                d = (en[sel] - st[sel]) / 100
                print("     learn block durations: median %.2f p90 %.2f max %.2f; slowest blocks %s" % (np.median(d), np.percentile(d, 90), d.max(), idx[sel][np.argsort(d)[-5:]] - 256))
k = htm.active_columns
for i in (3, 7):
    mm = t[i][:, 0] > 0
    if mm.any():
        st, en = t[i][mm, 0], t[i][mm, 1]
        idx = np.nonzero(mm)[0]
        for lo, hi, nm in ((256, 4096, "scan"), (0, 256, "close")):
            sel = (idx >= lo) & (idx < hi)
            if sel.any():
                d = (en[sel] - st[sel]) / 100
                print(f"  slot {i} {nm}: start {(st[sel].min() - st.min()) / 100:.2f}..{(st[sel].max() - st.min()) / 100:.2f} end {(en[sel].min() - st.min()) / 100:.2f}..{(en[sel].max() - st.min()) / 100:.2f}; block duration median {np.median(d):.2f} p90 {np.percentile(d, 90):.2f} max {d.max():.2f}")
        sc = (idx >= 256)
        order = np.argsort(st[sc])
        print("    scan blocks started by us:", [(int((st[sc] - st.min() <= x * 100).sum())) for x in (0.5, 1, 2, 4, 6, 8, 10, 12)])
        print("    scan blocks ended by us:", [(int((en[sc] - st.min() <= x * 100).sum())) for x in (2, 4, 6, 8, 10, 12, 14, 16, 18)])
