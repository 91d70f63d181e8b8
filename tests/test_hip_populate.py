"""GPU: pre-populated segment pools (htm_populate, BASELINE.json configs[4]: 255 segments per cell generated on the
device) against the oracle's twin generator, and the steps that follow -- unsharded and column-sharded."""

import numpy as np
import pytest

from oracle import HTMOracle, TMParams, TemporalMemoryOracle, canonical_synapses

pytestmark = pytest.mark.gpu


def _engine(C, K, k, tmp, capacity, slots=64, seed=0):
    import bithtm_amd as B
    from bithtm_amd.engine import Engine
    distal = B.PredictiveProjection(C * K, segment_capacity=capacity, segment_slots=slots,
                                    **{f: getattr(tmp, f) for f in tmp.__dataclass_fields__})
    return Engine(0, C, K, k, distal=distal, seed=seed)


def _compare_store(eng, ora, sample=None):
    st = eng.read_store()
    S = ora.S
    assert st["S"] == S
    assert np.array_equal(st["seg_cell"], ora.seg_cell[:S])
    assert np.array_equal(st["seg_nsyn"], ora.seg_nsyn[:S])
    assert np.array_equal(st["segcount"], ora.segcount)
    ids = np.arange(S) if sample is None else sample
    a = canonical_synapses(st["seg_cell"][ids], st["presyn"][ids], st["perm"][ids])
    b = canonical_synapses(ora.seg_cell[ids], ora.presyn[ids], ora.perm[ids])
    for s, (x, y) in zip(ids, zip(a, b)):
        assert np.array_equal(x[1], y[1]), f"segment {s}: presynaptic ids"
        assert np.array_equal(x[2].view(np.int32), y[2].view(np.int32)), f"segment {s}: permanence bits"


def test_populated_pool_equals_the_oracles_and_learns_like_it():
    """A small pool with low thresholds: the generated segments match, learn, get punished, die and are recycled."""
    C, K, k = 512, 8, 12
    tmp = TMParams(segment_activation_threshold=4, segment_matching_threshold=3, segment_sampling_synapses=16,
                   permanence_punishment=0.2)
    ora = TemporalMemoryOracle(C, K, tmp, seed=3)
    ora.populate(40, synapses=20, perm_lo=0.3, perm_hi=0.7, seed=11)
    eng = _engine(C, K, k, tmp, capacity=1 << 19, seed=3)
    eng.populate(40, synapses=20, perm_lo=0.3, perm_hi=0.7, seed=11)
    _compare_store(eng, ora)                        # (20 draws from 4 096 cells: the distinct-cells rule is exercised)
    rng = np.random.RandomState(4)
    seqs = [np.sort(rng.choice(C, k, replace=False)) for _ in range(7)]
    for t in range(40):
        cols = seqs[t % 7]
        want = ora.step(cols, learning=(t % 9) != 5)
        eng.tm_step(cols, learning=(t % 9) != 5)
        d = eng.read_distal()
        od = want.distal_state
        assert np.array_equal(d["matching_segment"], od.matching_segment), t
        assert np.array_equal(d["matching_segment_active"], od.matching_segment_active), t
        assert np.array_equal(d["segment_potential"], od.segment_potential), t
        assert np.array_equal(d["max_jittered_potential"].view(np.int32), od.max_jittered_potential.view(np.int32)), t
    eng.check_capacity()
    _compare_store(eng, ora)
    assert (ora.seg_nsyn[:ora.S] < 3).any() or ora.S > 40 * C * K      # deaths or growth happened


def test_configs4_pool_at_reduced_size_255_segments_per_cell():
    """BASELINE.json configs[4] at 2 048 columns x 16 cells: 255 segments per cell x 32 synapses, permanences
    U[0.3, 0.7), seed 0, generated on the device: 8.4 M segments, the large-pool scan."""
    C, K, k = 2048, 16, 41
    tmp = TMParams()
    ora = TemporalMemoryOracle(C, K, tmp, seed=0)
    ora.populate(255, synapses=32, seed=0)
    eng = _engine(C, K, k, tmp, capacity=ora.S + (1 << 16), seed=0)
    eng.populate(255, synapses=32, seed=0)
    info = eng.info()
    assert info.segments == ora.S == C * K * 255
    rng = np.random.RandomState(1)
    _compare_store_sample = rng.choice(ora.S, 4000, replace=False)
    st_nsyn = eng.read(11, np.int32, ora.S)         # HTM_F_SEG_NSYN
    assert np.array_equal(st_nsyn, ora.seg_nsyn[:ora.S])
    seqs = [np.sort(rng.choice(C, k, replace=False)) for _ in range(3)]
    for t in range(4):
        cols = seqs[t % 3]
        want = ora.step(cols)
        eng.tm_step(cols)
        od = want.distal_state
        d = eng.read_distal()
        assert np.array_equal(d["segment_potential"], od.segment_potential), t          # every segment's potential
        assert np.array_equal(d["matching_segment"], od.matching_segment), t
        assert np.array_equal(eng.read(6, np.uint32, C), _words(want.cell_prediction)), t   # HTM_F_CELL_PREDICTION
    info = eng.check_capacity()
    assert info.segments == ora.S                   # the bursting columns' new segments included
    st = eng.read_store()
    ids = np.concatenate([_compare_store_sample, np.arange(C * K * 255, ora.S)])
    a = canonical_synapses(st["seg_cell"][ids], st["presyn"][ids], st["perm"][ids])
    b = canonical_synapses(ora.seg_cell[ids], ora.presyn[ids], ora.perm[ids])
    assert all(x[0] == y[0] and np.array_equal(x[1], y[1]) and np.array_equal(x[2].view(np.int32), y[2].view(np.int32)) for x, y in zip(a, b))


def _words(mat):
    return (mat.astype(np.uint32) << np.arange(mat.shape[1], dtype=np.uint32)).sum(axis=1).astype(np.uint32)


@pytest.mark.parametrize("world", [2, 4])
def test_populated_pool_column_sharded(world):
    """The same generated pool on a column-sharded group (each rank generates the rows of its own cells), SP + TM with
    learning, against the unsharded oracle: the ranks hold exactly their own segments, under the global ids."""
    import bithtm_amd as B
    from bithtm_amd import _lib as L
    from bithtm_amd.distributed import LocalGroup
    from bithtm_amd.engine import words_to_bool
    I, C, K, P, seed = 128, 1024, 8, 9, 21
    k = round(C * 0.02)
    tmp = TMParams(segment_activation_threshold=4, segment_matching_threshold=3, segment_sampling_synapses=12,
                   permanence_punishment=0.25)
    np.random.seed(seed)
    perm = np.random.randn(C, I) * 0.1
    ora = HTMOracle(I, C, K, active_columns=k, seed=seed, tm_params=tmp, permanence=perm)
    ora.temporal_memory.populate(12, synapses=16, seed=5)

    def parts(r):
        prox = B.DenseProjection.__new__(B.DenseProjection)
        prox.input_dim, prox.output_dim = I, C
        prox.permanence_threshold, prox.permanence_increment, prox.permanence_decrement = 0.0, 0.03, 0.015
        prox._engine, prox._permanence = None, perm
        return dict(proximal=prox, boosting=B.ExponentialBoosting(C, k),
                    distal=B.PredictiveProjection(C * K, segment_capacity=1 << 18, segment_slots=64,
                                                  **{f: getattr(tmp, f) for f in tmp.__dataclass_fields__}))
    group = LocalGroup(world, I, C, K, active_columns=k, make_parts=parts, seed=seed)
    for m in group.members:
        m.engine.populate(12, synapses=16, seed=5)
    rng = np.random.RandomState(seed + 1)
    bank = rng.rand(P, I) < 0.1
    otm = ora.temporal_memory
    for t in range(60):
        x = bank[t % P] ^ (rng.rand(I) < 0.01)
        o_sp, o_tm = ora.step(x)
        group.process(x)
        for m in group.members:
            eng = m.engine
            c0, c1 = m.column_range
            info = eng.check_capacity()
            assert info.segments == otm.S, (t, m.rank)
            assert np.array_equal(eng.read(L.F_ACTIVE_COLUMN, np.int32, k), o_sp.active_column), (t, m.rank)
            pred = words_to_bool(eng.read(L.F_CELL_PREDICTION, np.uint32, C), K)
            assert np.array_equal(pred[c0:c1], o_tm.cell_prediction[c0:c1]), (t, m.rank)
            gid = eng.read(L.F_SEG_GID, np.int32, info.local_segments)
            live = np.flatnonzero(gid >= 0)
            owned = np.flatnonzero((otm.seg_cell[:otm.S] // K >= c0) & (otm.seg_cell[:otm.S] // K < c1))
            assert np.array_equal(np.sort(gid[live]), owned), (t, m.rank)
            nsyn = eng.read(L.F_SEG_NSYN, np.int32, info.local_segments)
            assert np.array_equal(nsyn[live], otm.seg_nsyn[gid[live]]), (t, m.rank)
    for m in group.members:
        eng = m.engine
        st = eng.read_store()
        live = np.flatnonzero(st["seg_gid"] >= 0)
        g = st["seg_gid"][live]
        a = canonical_synapses(st["seg_cell"][live], st["presyn"][live], st["perm"][live])
        b = canonical_synapses(otm.seg_cell[g], otm.presyn[g], otm.perm[g])
        assert all(x[0] == y[0] and np.array_equal(x[1], y[1]) and np.array_equal(x[2].view(np.int32), y[2].view(np.int32)) for x, y in zip(a, b))
