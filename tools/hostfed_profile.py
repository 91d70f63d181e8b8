import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
w = dict(bench.WORKLOAD)
noisy, perm = bench.make_inputs(w)
htm = bench.build_htm(w, perm, 0)
htm.run(noisy, 500)
eng = htm.engine
for t in range(100): eng.step(noisy[t % 1000])
eng.sync()
os.environ["X"]="1"
eng.profile(True)
for t in range(300): eng.step(noisy[(600+t) % 1000])
prof = eng.profile_read()
eng.profile(False)
print({n: round(1e3*ms/c,2) for n,(ms,c) in prof.items() if c})
