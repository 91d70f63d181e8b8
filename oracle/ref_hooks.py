"""Plug-ins that make the *unmodified* reference deterministic.  TEST INFRASTRUCTURE ONLY.

The reference exposes constructor hooks (`networks.py:16,22-24`: `boosting=`,
`inhibition=`) and draws its random numbers from the module-global `np.random.rand`.
This file supplies

  * `StableTopK`       -- an `inhibition=` object (interface of GlobalInhibition,
                          regularizations.py:24-29) implementing the documented tie rule;
  * `DocumentedExpBoosting` -- a `boosting=` object (interface of ExponentialBoosting,
                          regularizations.py:4-21) whose only difference is the exp;
  * `keyed_rand(...)`  -- a context manager replacing `np.random.rand` by the keyed
                          generator.  The replacement looks at its caller's frame to learn
                          what the draw is for (which cells / segments), because the
                          reference only passes a shape.

None of this is imported by the product.  It needs /root/reference on sys.path only for the
frame's function names; it contains no reference code.
"""

import contextlib
import sys

import numpy as np

from .fexp import exp_f32
from .htm_oracle import stable_topk, sp_derived, SPParams
from .keyed_rng import draw_unit, STREAM_LEAST_USED, STREAM_GROWTH, STREAM_SEGMENT_JITTER


class StableTopK:
    def __init__(self, active_outputs):
        self.active_outputs = active_outputs
        self.ambiguous_calls = 0
        self.calls = 0

    def process(self, input_activation):
        from .htm_oracle import topk_is_unambiguous
        self.calls += 1
        if not topk_is_unambiguous(input_activation, self.active_outputs):
            self.ambiguous_calls += 1
        return stable_topk(input_activation, self.active_outputs)


class DocumentedExpBoosting:
    def __init__(self, output_dim, active_outputs, intensity=0.3, momentum=0.99):
        self.density = active_outputs / output_dim
        self.intensity = intensity
        self.momentum = momentum
        self.duty_cycle = np.zeros(output_dim, dtype=np.float32)
        self._d = sp_derived(SPParams(boost_intensity=intensity, boost_momentum=momentum),
                             output_dim, active_outputs)

    def process(self, input_activation):
        factor = exp_f32(self._d.coef32 * self.duty_cycle)
        return factor * input_activation

    def update(self, active_input):
        self.duty_cycle *= self.momentum
        self.duty_cycle[active_input] += 1.0 - self.momentum


class KeyedRand:
    """Callable standing in for np.random.rand while the reference runs one TM step."""

    def __init__(self, seed, cell_dim, fallback):
        self.seed = seed
        self.cell_dim = cell_dim
        self.step = 0
        self.fallback = fallback
        self.log = []          # (step, stream, shape) of every keyed draw
        self.growth_ties = 0   # rows where a priority tie straddled the cut (see below)

    def __call__(self, *shape):
        frame = sys._getframe(1)
        name = frame.f_code.co_name
        loc = frame.f_locals
        if name == "evaluate_cell_least_used":            # networks.py:87
            cols = np.asarray(loc["relevant_column"], dtype=np.int64)
            flat = cols[:, None] * self.cell_dim + np.arange(self.cell_dim)
            out = draw_unit(self.seed, STREAM_LEAST_USED, self.step, flat)
            stream = STREAM_LEAST_USED
        elif name == "add_edge":                           # projections.py:120
            segs = np.asarray(loc["learning_output"], dtype=np.int64)
            winners = np.asarray(loc["winner_input"], dtype=np.int64)
            out = np.zeros((len(segs), len(winners) + 1), dtype=np.float64)
            if len(segs) and len(winners):
                out[:, :-1] = draw_unit(self.seed, STREAM_GROWTH, self.step, segs[:, None], winners[None, :])
            stream = STREAM_GROWTH
        elif name == "fill_jittered_potential_info":       # projections.py:235
            segs = np.asarray(loc["state"].matching_segment, dtype=np.int64)
            out = draw_unit(self.seed, STREAM_SEGMENT_JITTER, self.step, segs)
            stream = STREAM_SEGMENT_JITTER
        else:
            return self.fallback(*shape)
        assert out.shape == tuple(shape), (name, out.shape, shape)
        self.log.append((self.step, stream, tuple(shape)))
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


def import_reference(path="/root/reference"):
    """Import the reference package read-only (build container only)."""
    if path not in sys.path:
        sys.path.insert(0, path)
    import bithtm  # noqa: F401
    from bithtm import networks, projections, regularizations, reference_implementations
    return SimpleNamespaceLike(bithtm=bithtm, networks=networks, projections=projections,
                               regularizations=regularizations,
                               reference_implementations=reference_implementations)


class SimpleNamespaceLike:
    def __init__(self, **kw):
        self.__dict__.update(kw)
