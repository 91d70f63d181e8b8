"""CPU: the engine's host code under AddressSanitizer + UndefinedBehaviorSanitizer.

htm_engine.hip is compiled host-only (`hipcc --offload-host-only -fsanitize=address,undefined`: the kernels are parsed, not
emitted) and linked against tests/host_stub/hip_stub_runtime.cpp -- a HIP runtime made of host memory, in which every copy
the library makes is a bounds-checked memcpy and a kernel launch checks its grid and does nothing.  tests/host_stub/driver.py
then drives the C ABI through the Python classes in a child process that has the sanitizer runtime preloaded: whole models
in every call pattern, state import / export round trips (unsharded, 2 and 4 column shards, 32 and 64 cell slots per
column) compared with the arrays that went in, the L1 entry points, the error paths.  (GPU AddressSanitizer is not
available on this pool; this covers the 2 400 lines of host C++ that never ran under a sanitizer.)"""
import glob
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG_DIR = "/opt/rocm/lib/llvm"


def _build(tmp):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    clang = os.path.join(CLANG_DIR, "bin", "clang++")
    flags = ["-std=c++17", "-O1", "-g", "-fPIC", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
    obj, stub, lib = (os.path.join(tmp, n) for n in ("engine_host.o", "hip_stub.o", "libbithtm_host_san.so"))
    subprocess.run([hipcc, "--offload-host-only", "-ffp-contract=off", "-w"] + flags + ["-c", os.path.join(ROOT, "bithtm_amd", "csrc", "htm_engine.hip"), "-o", obj],
                   check=True, capture_output=True)
    subprocess.run([clang, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-w"] + flags + ["-c", os.path.join(ROOT, "tests", "host_stub", "hip_stub_runtime.cpp"), "-o", stub],
                   check=True, capture_output=True)
    undefined = subprocess.run(["nm", "-u", obj], check=True, capture_output=True, text=True).stdout
    fatbin = re.search(r"__hip_fatbin_\w+", undefined)       # (the device image the host object expects beside it: there is none)
    subprocess.run([clang, "-shared", "-fsanitize=address,undefined", "-shared-libsan", obj, stub, "-ldl", "-o", lib] +
                   ([f"-Wl,--defsym={fatbin.group(0)}=0"] if fatbin else []), check=True, capture_output=True)
    return lib


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_engine_host_code_under_address_and_ub_sanitizers(tmp_path):
    lib = _build(str(tmp_path))
    runtime = glob.glob(os.path.join(CLANG_DIR, "lib", "clang", "*", "lib", "linux", "libclang_rt.asan-x86_64.so"))
    assert runtime, "AddressSanitizer runtime not found"
    env = dict(os.environ, BITHTM_LIBRARY=lib, LD_PRELOAD=runtime[0], ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", BITHTM_EAGER_BELOW="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_stub", "driver.py")], env=env, capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0 and "host sanitizer driver: ok" in r.stdout, tail
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, tail
