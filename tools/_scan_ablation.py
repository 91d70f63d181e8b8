import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
w = dict(bench.WORKLOAD)
noisy, perm = bench.make_inputs(w)
os.environ.pop("BITHTM_DBG", None)
dbg = sys.argv[1]
htm = bench.build_htm(w, perm, 0)          # created without dbg
eng = htm.engine
bank = eng.upload_bank(noisy)
eng.run(bank, len(noisy), 1500)
st = htm.state_dict()
del htm, eng
os.environ["BITHTM_DBG"] = dbg
htm = bench.build_htm(w, np.zeros_like(st["sp_permanence"]), 0)
htm.load_state_dict(st)
eng = htm.engine
bank = eng.upload_bank(noisy)
eng.run(bank, len(noisy), 3, learning=False, use_graph=False, pipeline=False)
eng.profile(True)
eng.run(bank, len(noisy), 60, learning=False, use_graph=False, pipeline=False)
prof = eng.profile_read()
ms, n = prof["tm_scan"]
print(f"dbg={dbg}: tm_scan {1e3*ms/n:6.2f} us over {n} launches")
