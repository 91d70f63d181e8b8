"""Live differential test: unmodified reference (through its own hooks) vs. the oracle.

Runs only where /root/reference exists (the build container); skipped on the GPU box."""

import os

import pytest

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/bithtm"),
                                reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref():
    from oracle.ref_hooks import import_reference
    return import_reference()


def test_lockstep_small(ref):
    import refdiff
    stats, _, _ = refdiff.run_lockstep(ref, seed=31, input_dim=200, column_dim=1024, cell_dim=8, patterns=50,
                                       density=0.1, noise=0.02, steps=330, store_every=30)
    assert stats["segments"] > 1000


def test_lockstep_with_a_wide_epsilon(ref):
    """TemporalMemory.process(epsilon=0.3): best-matching and least-used ties a third of a count wide (several winner
    cells per bursting column, several best segments per cell)."""
    import refdiff
    stats, _, _ = refdiff.run_lockstep(ref, seed=33, input_dim=160, column_dim=1024, cell_dim=8, patterns=20,
                                       density=0.1, noise=0.02, steps=160, store_every=20, jump=0.1, epsilon=0.3)
    assert stats["segments"] > 500


def test_lockstep_with_earlier_states_as_prev_state(ref):
    """TemporalMemory.process(prev_state=): every seventh step runs from the state of three steps earlier."""
    import refdiff
    stats, _, _ = refdiff.run_lockstep(ref, seed=34, input_dim=160, column_dim=1024, cell_dim=8, patterns=20,
                                       density=0.1, noise=0.02, steps=150, store_every=15,
                                       prev_schedule=lambda t: 3 if t > 10 and t % 7 == 0 else None)
    assert stats["segments"] > 500


def test_lockstep_learning_off_and_jumps(ref):
    import refdiff
    stats, _, _ = refdiff.run_lockstep(ref, seed=32, input_dim=160, column_dim=1024, cell_dim=16, patterns=30,
                                       density=0.1, noise=0.02, steps=200, store_every=25, jump=0.2,
                                       learning_schedule=lambda t: (t % 17) != 3)
    assert stats["segments"] > 500


def test_the_products_bridge_makes_the_reference_agree(ref, monkeypatch):
    """bithtm_amd.reference_bridge.keyed_rand -- what INTEGRATION.md hands to a maintainer of the reference -- in place of the
    test infrastructure's own patch: the unmodified reference, drawing the engine's keyed numbers through it, stays in
    lock-step with the oracle (and so with the engine, which the GPU tests pin to the oracle)."""
    import numpy as np
    import refdiff
    from bithtm_amd.reference_bridge import keyed_rand, keyed_draws
    from oracle.keyed_rng import draw_unit
    monkeypatch.setattr(refdiff, "keyed_rand", keyed_rand)
    stats, _, _ = refdiff.run_lockstep(ref, seed=35, input_dim=160, column_dim=1024, cell_dim=8, patterns=20,
                                       density=0.1, noise=0.02, steps=120, store_every=20, jump=0.1)
    assert stats["segments"] > 300
    # ... and the C entry behind it gives the oracle's numbers
    a, b = np.arange(500, dtype=np.uint32) * 7919, np.arange(500, dtype=np.uint32)[::-1].copy()
    for stream in (1, 2, 3):
        assert np.array_equal(keyed_draws(35, stream, 99, a, b if stream == 2 else None), draw_unit(35, stream, 99, a, b if stream == 2 else 0))
