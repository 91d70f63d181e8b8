"""Build libbithtm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbithtm_hip.so")
SOURCES = ["htm_engine.hip"]
HEADERS = ["htm_dev.h", "htm_sp_kernels.h", "htm_tm_kernels.h", "htm_pipeline.h", "htm_rng.h", "htm_fexp.h",
           os.path.join("..", "..", "include", "bithtm_hip.h")]
# -ffp-contract=off: several kernels must round exactly like the NumPy expressions they replace
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-result", "-Wno-unused-value"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("BITHTM_EXTRA_FLAGS", "").split()      # diagnostic builds (e.g. -DBITHTM_LEARN_STAMPS)
    cmd = [hipcc] + FLAGS + extra + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout)
    if verbose and r.stdout.strip():
        print(r.stdout, file=sys.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
