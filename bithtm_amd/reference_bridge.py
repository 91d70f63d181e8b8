"""For a maintainer of the reference (cokwa/bitHTM) who wants to see their own installation and this engine agree bit for
bit: the three places where the reference's results are implementation-defined, each closed from the
reference's side with what the reference already offers -- constructor slots and the module-global `np.random.rand`.

    import bithtm, bithtm_amd
    from bithtm_amd.reference_bridge import keyed_rand

    ours = bithtm_amd.HierarchicalTemporalMemory(I, C, K, seed=7)
    ref_sp = bithtm.SpatialPooler(I, C, k,
                                  boosting=bithtm_amd.ExponentialBoosting(C, k),      # the documented exp (regularizations.py:16)
                                  inhibition=bithtm_amd.GlobalInhibition(k))           # the tie rule (regularizations.py:28-29)
    ref_sp.proximal_projection.permanence[:] = ours.spatial_pooler.proximal_projection.permanence
    ref_tm = bithtm.TemporalMemory(C, K)
    with keyed_rand(seed=7, cell_dim=K) as draws:           # np.random.rand -> the engine's keyed draws
        for x in inputs:
            sp_state = ref_sp.process(x)
            tm_state = ref_tm.process(sp_state)
            draws.step += 1
            ours.process(x)                                 # same winners, same cells, same permanence bits

The replacement for `np.random.rand` learns what a draw is for from its caller's frame -- the reference only passes a shape
-- by the NAMES of the reference's functions and locals (`evaluate_cell_least_used`: `relevant_column`; `add_edge`:
`learning_output`, `winner_input`; `fill_jittered_potential_info`: `state.matching_segment`); it contains none of the
reference's code.  The numbers are htm_keyed_draws' (include/bithtm_hip.h), computed here by its NumPy twin.  The engine
cannot go the other way and consume MT19937's draws: their shapes -- (learning segments, previous winners + 1), (matching
segments,) -- depend on data the timestep has not produced when it starts.
"""

import contextlib
import sys

import numpy as np

from ._keyed import draw_unit, STREAM_LEAST_USED, STREAM_GROWTH, STREAM_SEGMENT_JITTER


class KeyedRand:
    """Stands in for np.random.rand while the reference runs; `step` = the timestep index (the caller advances it)."""

    def __init__(self, seed, cell_dim, fallback):
        self.seed, self.cell_dim, self.fallback = seed, cell_dim, fallback
        self.step = 0
        self.calls = {STREAM_LEAST_USED: 0, STREAM_GROWTH: 0, STREAM_SEGMENT_JITTER: 0}

    def __call__(self, *shape):
        frame = sys._getframe(1)
        where, names = frame.f_code.co_name, frame.f_locals
        if where == "evaluate_cell_least_used":            # networks.py:87: one number per cell of the columns in question
            columns = np.asarray(names["relevant_column"], dtype=np.int64)
            out = draw_unit(self.seed, STREAM_LEAST_USED, self.step, columns[:, None] * self.cell_dim + np.arange(self.cell_dim))
            stream = STREAM_LEAST_USED
        elif where == "add_edge":                           # projections.py:120: (learning segments, previous winners + 1)
            segments = np.asarray(names["learning_output"], dtype=np.int64)
            winners = np.asarray(names["winner_input"], dtype=np.int64)
            out = np.zeros((len(segments), len(winners) + 1), dtype=np.float64)     # (the last column stands for "no cell": never drawn)
            if len(segments) and len(winners):
                out[:, :-1] = draw_unit(self.seed, STREAM_GROWTH, self.step, segments[:, None], winners[None, :])
            stream = STREAM_GROWTH
        elif where == "fill_jittered_potential_info":       # projections.py:235: one number per matching segment
            out = draw_unit(self.seed, STREAM_SEGMENT_JITTER, self.step, np.asarray(names["state"].matching_segment, dtype=np.int64))
            stream = STREAM_SEGMENT_JITTER
        else:
            return self.fallback(*shape)
        if out.shape != tuple(shape):
            raise RuntimeError(f"keyed_rand: {where} asked for {shape}, its arguments give {out.shape}: another version of the reference?")
        self.calls[stream] += 1
        return out


@contextlib.contextmanager
def keyed_rand(seed, cell_dim):
    original = np.random.rand
    patch = KeyedRand(seed, cell_dim, original)
    np.random.rand = patch
    try:
        yield patch
    finally:
        np.random.rand = original


def keyed_draws(seed, stream, step, a, b=None):
    """htm_keyed_draws through the library itself (the C entry a non-Python harness would call)."""
    import ctypes as C
    from . import _lib
    a = np.ascontiguousarray(a, dtype=np.uint32).ravel()
    bb = None if b is None else np.ascontiguousarray(b, dtype=np.uint32).ravel()
    out = np.empty(len(a), dtype=np.float64)
    rc = _lib.load().htm_keyed_draws(C.c_uint32(seed & 0xFFFFFFFF), int(stream), C.c_uint32(step & 0xFFFFFFFF), a.ctypes.data,
                                     None if bb is None else bb.ctypes.data, len(a), out.ctypes.data)
    if rc:
        raise ValueError("htm_keyed_draws: bad stream or arguments")
    return out
