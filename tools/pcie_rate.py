"""Timesteps/s when every step's input comes from host memory through htm_step (the reference's
calling convention, networks.py:146) with no State read-back -- the PCIe-inclusive figure quoted in
DESIGN.md; never bench.py's `value`."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
w = dict(bench.WORKLOAD)
noisy, perm = bench.make_inputs(w)
htm = bench.build_htm(w, perm, 0)
for t in range(500):
    htm.process(noisy[t % len(noisy)])
htm.engine.sync()
t0 = time.perf_counter()
n = 2000
for t in range(n):
    htm.process(noisy[(500 + t) % len(noisy)])
htm.engine.sync()
dt = time.perf_counter() - t0
print(f"host-fed htm.process(): {n / dt:.0f} timesteps/s ({1e6 * dt / n:.1f} us/step) incl. bit-packing, ctypes and H2D of {w['input_dim'] // 8} B per step")
from bithtm_amd.engine import pack_bits
packed = [pack_bits(x, htm.engine.words).copy() for x in noisy]
htm.engine.sync()
t0 = time.perf_counter()
for t in range(n):
    htm.process(packed[(2500 + t) % len(packed)])
htm.engine.sync()
dt = time.perf_counter() - t0
print(f"the same with inputs the caller keeps packed (uint32[{htm.engine.words}]): {n / dt:.0f} timesteps/s ({1e6 * dt / n:.1f} us/step)")
eng = htm.engine
t0 = time.perf_counter()
for t in range(n):
    eng.step(packed[(4500 + t) % len(packed)])
eng.sync()
dt = time.perf_counter() - t0
print(f"engine.step() alone (one ctypes call per timestep, no State objects): {n / dt:.0f} timesteps/s ({1e6 * dt / n:.1f} us/step)")
