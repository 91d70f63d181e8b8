"""Build libbithtm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""

import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbithtm_hip.so")
RESOURCES = os.path.join(HERE, "libbithtm_hip.resources.json")     # what the compiler gave every kernel (written beside the library)
SOURCES = ["htm_engine.hip"]
HEADERS = ["htm_dev.h", "htm_sp_kernels.h", "htm_tm_kernels.h", "htm_pipeline.h", "htm_rng.h", "htm_fexp.h",
           os.path.join("..", "..", "include", "bithtm_hip.h")]
# -ffp-contract=off: several kernels must round exactly like the NumPy expressions they replace
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-result", "-Wno-unused-value",
         "-Rpass-analysis=kernel-resource-usage"]


def _stale():
    if not os.path.exists(LIB) or not os.path.exists(RESOURCES):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def _kernel_resources(log):
    """The compiler's per-kernel remarks -> {mangled name: {"vgprs", "agprs", "sgprs", "scratch_bytes_per_lane", "occupancy",
    "lds_bytes"}}.  k_learn_scan_emit lives at 80 VGPRs (6 waves per SIMD) with NO scratch: a single spilled register slows
    every role of that launch by microseconds (LABNOTES.md) -- tests/test_host_and_abi.py holds the build to it."""
    fields = {"VGPRs": "vgprs", "AGPRs": "agprs", "SGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
              "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds_bytes"}
    out, cur = {}, None
    for line in log.splitlines():
        m = re.search(r"remark: +([^:]+): (\S+) \[-Rpass-analysis=kernel-resource-usage\]", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == "Function Name":
            cur = out.setdefault(val, {})
        elif cur is not None and key in fields:
            cur[fields[key]] = int(val)
    return out


def _without_remarks(log):
    """The compiler's output minus the resource remarks (each with its `In file included from` chain and source snippet)."""
    out, pending, in_remark = [], [], False
    for line in log.splitlines():
        if "kernel-resource-usage" in line:
            pending, in_remark = [], True
        elif line.startswith("In file included from"):
            pending.append(line)
            in_remark = False
        elif in_remark and re.match(r"\s*\d*\s*\|", line):
            continue
        else:
            out += pending + [line]
            pending, in_remark = [], False
    return "\n".join(out)


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
    with open(RESOURCES, "w") as f:
        json.dump(_kernel_resources(r.stdout), f, indent=0, sort_keys=True)
    if verbose:
        rest = _without_remarks(r.stdout)
        if rest.strip():
            print(rest, file=sys.stderr)
    return LIB


def kernel_resources():
    """{mangled kernel name: resources} of the library as last built here (None if it was built elsewhere)."""
    build()
    if not os.path.exists(RESOURCES):
        return None
    with open(RESOURCES) as f:
        return json.load(f)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
