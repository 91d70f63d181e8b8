"""GPU: plug-in objects that live on the host and the L1 methods called on their own.

`SpatialPooler(..., proximal_projection=, boosting=, inhibition=)` (networks.py:16,22-24) accepts any object with the
reference's `process` / `update` methods.  The objects injected here are the reference-shaped test doubles of
oracle/ref_hooks.py (the ones the golden recorder hands to the unmodified reference) and a NumPy DenseProjection
written here; since they implement the documented policies, every combination must reproduce the oracle bit for bit."""

import numpy as np
import pytest

from oracle import HTMOracle, SpatialPoolerOracle
from oracle.ref_hooks import DocumentedExpBoosting, StableTopK

pytestmark = pytest.mark.gpu


class HostDenseProjection:
    """projections.py:6-24 in NumPy: a user's own proximal projection."""

    def __init__(self, permanence, threshold=0.0, increment=0.03, decrement=0.015):
        self.permanence = permanence.copy()
        self.permanence_threshold, self.permanence_increment, self.permanence_decrement = threshold, increment, decrement
        self.calls = 0

    def process(self, input_activation):
        self.calls += 1
        return ((self.permanence >= self.permanence_threshold) & input_activation).sum(axis=1)

    def update(self, input_activation, learning_output):
        self.permanence[learning_output] += input_activation * (self.permanence_increment + self.permanence_decrement) - self.permanence_decrement


@pytest.mark.parametrize("foreign", ["inhibition", "boosting", "proximal", "boosting+inhibition", "all"])
def test_htm_with_host_side_plugins_equals_the_oracle(foreign):
    import bithtm_amd as B
    I, C, K, k, seed = 200, 2048, 8, 41, 61
    np.random.seed(seed)
    perm = np.random.randn(C, I) * 0.1
    ora = HTMOracle(I, C, K, active_columns=k, seed=seed, permanence=perm)
    kw = {}
    if "inhibition" in foreign or foreign == "all":
        kw["inhibition"] = StableTopK(k)
    if "boosting" in foreign or foreign == "all":
        kw["boosting"] = DocumentedExpBoosting(C, k)
    if foreign in ("proximal", "all"):
        kw["proximal_projection"] = HostDenseProjection(perm)
    else:
        prox = B.DenseProjection(I, C)
        prox.permanence = perm
        kw["proximal_projection"] = prox
    sp = B.SpatialPooler(I, C, k, **kw)
    htm = B.HierarchicalTemporalMemory(I, C, K, active_columns=k, spatial_pooler=sp, temporal_memory=B.TemporalMemory(C, K, seed=seed))
    rng = np.random.RandomState(seed + 1)
    bank = rng.rand(12, I) < 0.1
    for t in range(90):
        x = bank[t % 12] ^ (rng.rand(I) < 0.01)
        learning = t % 13 != 7
        o_sp, o_tm = ora.step(x, learning=learning)
        s, m = htm.process(x, learning=learning)
        assert np.array_equal(np.sort(s.active_column), o_sp.active_column), t
        assert np.array_equal(np.asarray(s.overlaps), o_sp.overlaps), t
        assert np.array_equal(np.asarray(s.boosted_overlaps, dtype=np.float64).view(np.int64), o_sp.boosted_overlaps.view(np.int64)), t
        assert np.array_equal(m.cell_activation, o_tm.cell_activation) and np.array_equal(m.cell_prediction, o_tm.cell_prediction), t
        assert np.array_equal(np.sort(m.winner_cell[0] * K + m.winner_cell[1]), o_tm.winner_cell[0] * K + o_tm.winner_cell[1]), t
        assert np.array_equal(m.distal_state.matching_segment, o_tm.distal_state.matching_segment), t
    # the state each side keeps
    duty = kw["boosting"].duty_cycle if "boosting" in kw else sp.boosting.duty_cycle
    assert np.array_equal(np.asarray(duty, dtype=np.float32).view(np.int32), ora.spatial_pooler.duty_cycle.view(np.int32))
    final_perm = kw["proximal_projection"].permanence
    assert np.array_equal(np.asarray(final_perm).view(np.int64), ora.spatial_pooler.permanence.view(np.int64))
    if "inhibition" in kw:
        assert kw["inhibition"].calls == 90
    htm.engine.check_capacity()


def test_spatial_pooler_alone_with_a_host_side_inhibition():
    import bithtm_amd as B
    I, C, k = 333, 3000, 60
    np.random.seed(2)
    inh = StableTopK(k)
    sp = B.SpatialPooler(I, C, k, inhibition=inh)
    ora = SpatialPoolerOracle(I, C, k, permanence=sp.proximal_projection.permanence.copy())
    rng = np.random.RandomState(3)
    for t in range(60):
        x = rng.rand(I) < 0.1
        got, want = sp.process(x, learning=t % 5 != 2), ora.step(x, learning=t % 5 != 2)
        assert np.array_equal(got.active_column, want.active_column) and np.array_equal(got.overlaps, want.overlaps), t
    assert inh.calls == 60
    assert np.array_equal(sp.proximal_projection.permanence.view(np.int64), ora.permanence.view(np.int64))
    assert np.array_equal(sp.boosting.duty_cycle.view(np.int32), ora.duty_cycle.view(np.int32))


def test_l1_methods_on_their_own():
    """DenseProjection.process / update (projections.py:18-24), ExponentialBoosting.process / update
    (regularizations.py:15-21) and GlobalInhibition.process (:28-29) called directly, as the reference allows."""
    import bithtm_amd as B
    I, C, k = 150, 1000, 20
    np.random.seed(4)
    prox = B.DenseProjection(I, C, permanence_threshold=0.01)
    boost = B.ExponentialBoosting(C, k, intensity=0.5)
    inh = B.GlobalInhibition(k)
    from oracle import SPParams
    ora = SpatialPoolerOracle(I, C, k, params=SPParams(permanence_threshold=0.01, boost_intensity=0.5), permanence=prox.permanence.copy())
    rng = np.random.RandomState(5)
    for t in range(25):                                              # networks.py:26-35, spelled out by the caller
        x = rng.rand(I) < 0.15
        overlaps = prox.process(x)
        boosted = boost.process(overlaps)
        active = inh.process(boosted)
        prox.update(x, active)
        boost.update(active)
        want = ora.step(x)
        assert overlaps.dtype == np.int64 and np.array_equal(overlaps, want.overlaps), t
        assert np.array_equal(boosted.view(np.int64), want.boosted_overlaps.view(np.int64)), t
        assert np.array_equal(active, want.active_column), t
    assert np.array_equal(prox.permanence.view(np.int64), ora.permanence.view(np.int64))
    assert np.array_equal(boost.duty_cycle.view(np.int32), ora.duty_cycle.view(np.int32))


def test_standalone_temporal_memory_takes_more_columns_later():
    """The reference accepts any number of active columns per call; the engine of a stand-alone TemporalMemory is sized
    by the first call and replaced by a larger one, state and all, when more arrive."""
    import bithtm_amd as B
    from types import SimpleNamespace
    from oracle import TemporalMemoryOracle
    C, K = 512, 8
    tm = B.TemporalMemory(C, K, seed=3)
    ora = TemporalMemoryOracle(C, K, seed=3)
    rng = np.random.RandomState(6)
    seqs = [np.sort(rng.choice(C, n, replace=False)) for n in (16, 16, 16, 40, 16, 90, 40)]
    for t in range(70):
        cols = seqs[t % len(seqs)] if t >= 12 else seqs[t % 3]
        got, want = tm.process(SimpleNamespace(active_column=cols)), ora.step(cols)
        assert np.array_equal(got.cell_prediction, want.cell_prediction) and np.array_equal(got.cell_activation, want.cell_activation), t
        assert np.array_equal(got.winner_cell[0], want.winner_cell[0]) and np.array_equal(got.winner_cell[1], want.winner_cell[1]), t


def test_l1_methods_on_the_objects_of_a_live_pooler_do_not_disturb_its_steps():
    """DenseProjection.process / ExponentialBoosting.process called on their own on the objects BOUND to a live
    SpatialPooler / HierarchicalTemporalMemory use that pooler's engine (one phase of the current, unfinished timestep:
    keys, boosted overlaps, the top-digit histogram).  The whole steps that follow -- sp.process(), htm.process(), htm.run()
    -- must start over from a clean histogram and return the completed step's fields: interleaved here, against the oracle."""
    import bithtm_amd as B
    I, C, K, k, seed = 300, 2048, 8, 41, 71
    np.random.seed(seed)
    perm = np.random.randn(C, I) * 0.1
    rng = np.random.RandomState(seed + 1)
    bank = rng.rand(10, I) < 0.1
    # a Spatial Pooler on its own
    ora = SpatialPoolerOracle(I, C, k, permanence=perm)
    prox = B.DenseProjection(I, C)
    prox.permanence = perm
    sp = B.SpatialPooler(I, C, k, proximal_projection=prox)
    for t in range(40):
        x = bank[t % 10] ^ (rng.rand(I) < 0.01)
        if t % 3 == 1:
            probe = bank[(t + 5) % 10]
            assert np.array_equal(sp.proximal_projection.process(probe), ora.overlaps(probe)), t
        if t % 4 == 2:
            assert np.array_equal(sp.boosting.process(ora.overlaps(x)).view(np.int64), ora.boost(ora.overlaps(x)).view(np.int64)), t
        want, got = ora.step(x), sp.process(x)
        assert np.array_equal(got.active_column, want.active_column), t
        assert np.array_equal(got.overlaps, want.overlaps), t
        assert np.array_equal(got.boosted_overlaps.view(np.int64), want.boosted_overlaps.view(np.int64)), t
    # the same inside a HierarchicalTemporalMemory, with batched runs in between
    ora = HTMOracle(I, C, K, active_columns=k, seed=seed, permanence=perm)
    prox = B.DenseProjection(I, C)
    prox.permanence = perm
    htm = B.HierarchicalTemporalMemory(I, C, K, active_columns=k, spatial_pooler=B.SpatialPooler(I, C, k, proximal_projection=prox),
                                       temporal_memory=B.TemporalMemory(C, K, seed=seed))
    t = 0
    for rnd in range(12):
        probe = bank[(rnd + 3) % 10]
        assert np.array_equal(htm.spatial_pooler.proximal_projection.process(probe), ora.spatial_pooler.overlaps(probe)), rnd
        if rnd % 2:
            for _ in range(5):
                o_sp, o_tm = ora.step(bank[t % 10])
                t += 1
            htm.run(bank, 5)
            assert np.array_equal(htm.engine.read_sp_fields()["active_column"], o_sp.active_column), rnd
            assert np.array_equal(htm.engine.read_sp_fields()["overlaps"], o_sp.overlaps), rnd
        else:
            o_sp, o_tm = ora.step(bank[t % 10])
            s, m = htm.process(bank[t % 10])
            t += 1
            assert np.array_equal(s.active_column, o_sp.active_column), rnd
            assert np.array_equal(s.overlaps, o_sp.overlaps), rnd
            assert np.array_equal(m.cell_prediction, o_tm.cell_prediction), rnd
