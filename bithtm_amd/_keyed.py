"""Keyed random draws on the host: the NumPy twin of csrc/htm_rng.h (a stateless 32-bit hash -- "lowbias32" -- of
(seed, stream, step, what the number is for); the top 24 bits make a number in [0, 1) that is exact in float32).

Used where the timestep is orchestrated on the host because a plug-in object lives there (TemporalMemory with a foreign
`distal_projection=`): the "least used cell" jitter of networks.py:87 is then drawn here, with the device's numbers."""

import numpy as np

STREAM_LEAST_USED, STREAM_GROWTH, STREAM_SEGMENT_JITTER = 1, 2, 3

_M1, _M2 = np.uint32(0x7FEB352D), np.uint32(0x846CA68B)


def _mix32(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    x ^= x >> np.uint32(16)
    x *= _M1
    x ^= x >> np.uint32(15)
    x *= _M2
    x ^= x >> np.uint32(16)
    return x


def draw_unit(seed, stream, step, a, b=0):
    """htm_draw24(htm_stream_base(seed, stream, step), a, b) * 2**-24 as float64; `a`, `b` broadcast."""
    with np.errstate(over="ignore"):
        base = _mix32(np.uint32((int(seed) + int(stream) * 0x9E3779B9) & 0xFFFFFFFF))
        base = _mix32(base ^ np.uint32(int(step) & 0xFFFFFFFF))
        h = _mix32(_mix32(base ^ np.asarray(a).astype(np.uint32)) ^ np.asarray(b).astype(np.uint32))
    return (h >> np.uint32(8)).astype(np.float64) * (1.0 / 16777216.0)
