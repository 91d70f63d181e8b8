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


@pytest.mark.parametrize("foreign,K", [("inhibition", 8), ("boosting", 8), ("proximal", 8), ("boosting+inhibition", 8), ("all", 8), ("inhibition", 40), ("all", 64)])
def test_htm_with_host_side_plugins_equals_the_oracle(foreign, K):
    import bithtm_amd as B
    I, C, k, seed = 200, 2048, 41, 61
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


# ---- distal_projection= / temporal_memory= / spatial_pooler= (networks.py:50,55,134,143-144) ------------------------------
class OracleProjection:
    """A user's own PredictiveProjection with the reference's interface (projections.py:194-293), living on the host: the
    oracle's learning and scan behind `update` / `process` / `get_jittered_potential_info` / `bundle_segments`."""

    def __init__(self, column_dim, cell_dim, seed, params=None):
        from oracle import TMParams, TemporalMemoryOracle
        self.o = TemporalMemoryOracle(column_dim, cell_dim, params or TMParams(), seed)
        self.segment_matching_threshold = self.o.params.segment_matching_threshold
        self.C, self.K = column_dim, cell_dim
        self.calls = dict(process=0, update=0)

    @property
    def bundle_segments(self):
        return self.o.segcount

    def get_jittered_potential_info(self, state, matching_segment_bundle=None):
        return state.max_jittered_potential, state.matching_segment_jittered_potential

    def process(self, active_input, return_jittered_potential_info=True):
        self.calls["process"] += 1
        act = np.zeros(self.C * self.K, dtype=np.bool_)
        act[active_input] = True
        d = self.o._scan(act.reshape(self.C, self.K), self.o.step_index)
        self.o.step_index += 1
        return d

    def update(self, prev_state, input_activation, learning_output, output_punishment, winner_input=None, output_learning=None, epsilon=1e-8):
        if prev_state is None:
            return
        self.calls["update"] += 1
        o = self.o
        o.prev_distal, o.prev_activation, o.prev_winner = prev_state, np.asarray(input_activation).reshape(self.C, self.K), winner_input
        active_column = np.flatnonzero(~np.asarray(output_punishment).reshape(self.C, self.K).any(axis=1))
        o._learn(np.asarray(learning_output, dtype=np.int64), active_column, o.step_index)


def _tm_states_equal(t, got, want, K):
    assert np.array_equal(got.cell_activation, want.cell_activation), t
    assert np.array_equal(got.cell_prediction, want.cell_prediction), t
    assert np.array_equal(got.active_column_bursting[:, 0], want.active_column_bursting[:, 0]), t
    assert np.array_equal(got.winner_cell[0] * K + got.winner_cell[1], want.winner_cell[0] * K + want.winner_cell[1]), t
    assert np.array_equal(got.active_cell[0] * K + got.active_cell[1], want.active_cell[0] * K + want.active_cell[1]), t
    gd, wd = got.distal_state, want.distal_state
    assert np.array_equal(gd.matching_segment, wd.matching_segment), t
    assert np.array_equal(gd.segment_potential, wd.segment_potential), t
    assert np.array_equal(gd.matching_segment_activation, wd.matching_segment_activation), t
    assert np.array_equal(gd.matching_segment_active, wd.matching_segment_active), t
    assert np.array_equal(np.asarray(gd.max_jittered_potential).view(np.int32), wd.max_jittered_potential.view(np.int32)), t
    assert np.array_equal(np.asarray(gd.prediction), wd.prediction), t


def _column_sequences(C, k, n, seed):
    rng = np.random.RandomState(seed)
    return [np.sort(rng.choice(C, k, replace=False)) for _ in range(n)]


def test_temporal_memory_with_a_host_side_distal_projection():
    """TemporalMemory(distal_projection=<any object with the reference's interface>): TemporalMemory.process runs on the host
    (networks.py:91-128) and calls the user's update / process / get_jittered_potential_info where the reference does."""
    import bithtm_amd as B
    from types import SimpleNamespace
    from oracle import TemporalMemoryOracle
    C, K, k, seed = 512, 8, 20, 81
    proj = OracleProjection(C, K, seed)
    tm = B.TemporalMemory(C, K, distal_projection=proj, seed=seed)
    ora = TemporalMemoryOracle(C, K, seed=seed)
    seqs = _column_sequences(C, k, 9, 82)
    rng = np.random.RandomState(83)
    for t in range(80):
        cols = seqs[t % 9]
        learning = t % 11 != 6
        want = ora.step(cols, learning=learning)
        shuffled = cols[rng.permutation(k)]                         # (any order: per-column results come back in the caller's)
        got = tm.process(SimpleNamespace(active_column=shuffled), learning=learning)
        assert np.array_equal(got.active_column_bursting[np.argsort(shuffled, kind="stable"), 0], want.active_column_bursting[:, 0]), t
        got.active_column_bursting = got.active_column_bursting[np.argsort(shuffled, kind="stable")]
        _tm_states_equal(t, got, want, K)
        assert tm.last_state is got
    assert proj.calls["process"] == 80 and proj.calls["update"] > 60
    assert np.array_equal(proj.o.seg_nsyn[:proj.o.S], ora.seg_nsyn[:ora.S]) and proj.o.S == ora.S > 100


@pytest.mark.parametrize("default_mask", [False, True], ids=["mask-from-caller", "mask-built-by-the-library"])
def test_predictive_projection_methods_on_their_own_drive_the_device(default_mask):
    """PredictiveProjection.process / .update / .get_jittered_potential_info (projections.py:229-293) called on their own: a
    subclass of the device's projection is not the fused kind, so TemporalMemory orchestrates on the host and every call
    lands on the device through htm_tm_update / htm_tm_scan -- learning, allocation, recycling, scan -- against the oracle.
    default_mask: htm_tm_update's punish_words == NULL ("every cell of a column not listed", include/bithtm_hip.h) in place
    of the mask TemporalMemory builds (networks.py:107-108,111) -- the same cells, since every active column has a learning
    cell."""
    import bithtm_amd as B
    from types import SimpleNamespace
    from oracle import TMParams, TemporalMemoryOracle, canonical_synapses

    class MyProjection(B.PredictiveProjection):
        def update(self, prev_state, input_activation, learning_output, output_punishment, *a, **kw):
            return super().update(prev_state, input_activation, learning_output, None if default_mask else output_punishment, *a, **kw)

    C, K, k, seed = 1024, 16, 21, 91
    tmp = TMParams(segment_activation_threshold=9, segment_matching_threshold=7, segment_sampling_synapses=18, permanence_punishment=0.15,
                   permanence_decrement=0.12)
    proj = MyProjection(C * K, segment_slots=64, **{f: getattr(tmp, f) for f in tmp.__dataclass_fields__})
    tm = B.TemporalMemory(C, K, distal_projection=proj, seed=seed)
    assert not tm._own_distal
    ora = TemporalMemoryOracle(C, K, tmp, seed=seed)
    seqs = _column_sequences(C, k, 11, 92)
    rng = np.random.RandomState(93)
    for t in range(120):
        cols = seqs[int(rng.randint(11))] if rng.rand() < 0.15 else seqs[t % 11]
        learning = t % 13 != 5
        want = ora.step(cols, learning=learning)
        got = tm.process(SimpleNamespace(active_column=cols), learning=learning)
        _tm_states_equal(t, got, want, K)
    eng = proj._engine
    eng.check_capacity()
    st = eng.read_store()
    S = ora.S
    assert st["S"] == S and np.array_equal(st["seg_cell"], ora.seg_cell[:S]) and np.array_equal(st["seg_nsyn"], ora.seg_nsyn[:S])
    assert np.array_equal(st["segcount"], ora.segcount)
    a, b = canonical_synapses(st["seg_cell"], st["presyn"], st["perm"]), canonical_synapses(ora.seg_cell[:S], ora.presyn[:S], ora.perm[:S])
    assert all(np.array_equal(x[1], y[1]) and np.array_equal(x[2].view(np.int32), y[2].view(np.int32)) for x, y in zip(a, b))
    assert (ora.seg_nsyn[:S] < 7).any()               # segments died: recycling was exercised
    # the methods by hand, with an earlier State (written back as the device's previous step)
    d1 = proj.process(np.flatnonzero(want.cell_activation.reshape(-1)))
    assert np.array_equal(d1.matching_segment, ora._scan(want.cell_activation, ora.step_index).matching_segment)


@pytest.mark.parametrize("foreign", ["temporal_memory", "spatial_pooler", "both"])
def test_htm_with_foreign_layers(foreign):
    """HierarchicalTemporalMemory(spatial_pooler=, temporal_memory=) with objects that are not the device's own
    (networks.py:134,143-149; example.py:7-12 swaps the Temporal Memory): the two `process` calls of the reference, the
    device-backed layer stepping an engine of its own."""
    import bithtm_amd as B
    from types import SimpleNamespace
    from oracle import TemporalMemoryOracle
    I, C, K, k, seed = 200, 2048, 8, 41, 97
    np.random.seed(seed)
    perm = np.random.randn(C, I) * 0.1
    ora = HTMOracle(I, C, K, active_columns=k, seed=seed, permanence=perm)

    class UserTM:
        def __init__(self):
            self.o = TemporalMemoryOracle(C, K, seed=seed)

        def process(self, sp_state, learning=True):
            return self.o.step(np.sort(np.asarray(sp_state.active_column)), learning=learning)

    class UserSP:
        def __init__(self):
            self.o = SpatialPoolerOracle(I, C, k, permanence=perm)

        def process(self, input, learning=True):
            return self.o.step(input, learning=learning)

    kw = {}
    if foreign in ("temporal_memory", "both"):
        kw["temporal_memory"] = UserTM()
    else:
        kw["temporal_memory"] = B.TemporalMemory(C, K, seed=seed)
    if foreign in ("spatial_pooler", "both"):
        kw["spatial_pooler"] = UserSP()
    else:
        prox = B.DenseProjection(I, C)
        prox.permanence = perm
        kw["spatial_pooler"] = B.SpatialPooler(I, C, k, proximal_projection=prox)
    htm = B.HierarchicalTemporalMemory(I, C, K, active_columns=k, **kw)
    rng = np.random.RandomState(seed + 1)
    bank = rng.rand(10, I) < 0.1
    for t in range(70):
        x = bank[t % 10] ^ (rng.rand(I) < 0.01)
        o_sp, o_tm = ora.step(x)
        s, m = htm.process(x)
        assert np.array_equal(np.sort(s.active_column), o_sp.active_column), t
        assert np.array_equal(m.cell_prediction, o_tm.cell_prediction), t
        assert np.array_equal(m.winner_cell[0] * K + m.winner_cell[1], o_tm.winner_cell[0] * K + o_tm.winner_cell[1]), t
    with pytest.raises(RuntimeError):
        htm.run(bank, 3)
