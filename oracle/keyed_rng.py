"""Keyed (counter-based) random draws shared by the oracle and the HIP kernels.

The reference consumes the process-global MT19937 stream three times per timestep
(SURVEY.md §8c): `networks.py:87` (least-used-cell jitter, shape (k, K)),
`projections.py:120` (synapse-growth priority, shape (n_learn, n_prev_winner + 1)) and
`projections.py:235` (matching-segment jitter, shape (n_matching,)).  A sequential stream
cannot be reproduced by thousands of GPU lanes, so both this oracle and the kernels draw
every number from a stateless hash of *what the number is for*:

    draw24(seed, stream, step, a, b)  ->  integer m in [0, 2**24),   u = m * 2**-24

    stream 1  least-used jitter    a = flat cell id (col * K + cell)        b = 0
    stream 2  growth priority      a = segment id                           b = presynaptic
                                                                               flat cell id
    stream 3  segment jitter       a = segment id                           b = 0

`u` has 24 significant bits, so it is exact in float32 and float64 and `u < 1.0` always.
The golden-vector generator feeds the very same numbers to the unmodified reference by
patching `np.random.rand` (oracle/ref_hooks.py), so reference, oracle and GPU all see one
set of draws.

Mixer: the public-domain "lowbias32" integer finaliser, applied once per key word.
bithtm_amd/csrc/htm_rng.h is the device twin of this file.
"""

import numpy as np

STREAM_LEAST_USED = 1
STREAM_GROWTH = 2
STREAM_SEGMENT_JITTER = 3
STREAM_POPULATE_CELL = 4      # pre-populated pools (populate): a = segment id, b = synapse index
STREAM_POPULATE_PERM = 5

_M1 = np.uint32(0x7FEB352D)
_M2 = np.uint32(0x846CA68B)
_GOLD = 0x9E3779B9


def mix32(x):
    """lowbias32 on a uint32 ndarray (wrapping arithmetic)."""
    x = np.asarray(x, dtype=np.uint32).copy()
    x ^= x >> np.uint32(16)
    x *= _M1
    x ^= x >> np.uint32(15)
    x *= _M2
    x ^= x >> np.uint32(16)
    return x


def stream_base(seed, stream, step):
    """Per-(seed, stream, step) prefix of the hash, a uint32 scalar array."""
    with np.errstate(over="ignore"):
        h = mix32(np.uint32((int(seed) + int(stream) * _GOLD) & 0xFFFFFFFF))
        h = mix32(h ^ np.uint32(int(step) & 0xFFFFFFFF))
    return h


def draw24(seed, stream, step, a, b=0):
    """24-bit keyed draw; `a`, `b` broadcast against each other. Returns uint32 ndarray."""
    with np.errstate(over="ignore"):
        h0 = stream_base(seed, stream, step)
        a = np.asarray(a).astype(np.uint32)
        b = np.asarray(b).astype(np.uint32)
        h = mix32(h0 ^ a)
        h = mix32(h ^ b)
    return h >> np.uint32(8)


def draw32(seed, stream, step, a, b=0):
    """The same hash, all 32 bits (device twin: htm_draw32)."""
    with np.errstate(over="ignore"):
        h0 = stream_base(seed, stream, step)
        a = np.asarray(a).astype(np.uint32)
        b = np.asarray(b).astype(np.uint32)
        return mix32(mix32(h0 ^ a) ^ b)


def draw_unit(seed, stream, step, a, b=0):
    """The same draw as a float64 in [0, 1) with 24 significant bits."""
    return draw24(seed, stream, step, a, b).astype(np.float64) * (1.0 / 16777216.0)
