"""Lock-step comparison helper: unmodified reference (through its hooks) vs. the oracle.

Used by tests/test_oracle_vs_reference.py (live, build container only) and by
tests/golden/generate_golden.py (to record golden vectors).  Not collected by pytest.
"""

import numpy as np

from oracle import HTMOracle, canonical_synapses
from oracle.ref_hooks import StableTopK, DocumentedExpBoosting, keyed_rand, import_reference


def make_inputs(seed, patterns, input_dim, density):
    rng = np.random.RandomState(seed)
    return rng.rand(patterns, input_dim) < density, rng


def reference_store(ref_tm):
    """Reference synapse store in the oracle's canonical form."""
    dp = ref_tm.distal_projection
    sp = dp.segment_projection
    edge = sp.output_edge[:]
    perm = sp.output_permanence[:]
    presyn = np.where(edge == sp.invalid_output_edge, -1, sp.get_output_edge_target(edge)).astype(np.int64)
    seg_cell = dp.segment_bundle[:].squeeze(1) if len(dp.segment_bundle) else np.zeros(0, np.int32)
    return seg_cell, presyn, perm, sp.output_edges[:].squeeze(1), dp.bundle_segments


def compare_step(t, ref_sp, ref_tm, ora_sp, ora_tm, K):
    """Raise AssertionError on the first field that differs."""
    def eq(name, a, b):
        a, b = np.asarray(a), np.asarray(b)
        assert a.shape == b.shape, f"step {t}: {name} shape {a.shape} vs {b.shape}"
        assert np.array_equal(a, b), f"step {t}: {name} differs at {np.flatnonzero((a != b).reshape(-1))[:8]}"

    eq("active_column", np.sort(ref_sp.active_column), ora_sp.active_column)
    eq("active_column order", ref_sp.active_column, ora_sp.active_column)
    eq("overlaps", ref_sp.overlaps, ora_sp.overlaps)
    eq("boosted", ref_sp.boosted_overlaps, ora_sp.boosted_overlaps)
    eq("bursting", ref_tm.active_column_bursting, ora_tm.active_column_bursting)
    eq("cell_activation", ref_tm.cell_activation, ora_tm.cell_activation)
    eq("cell_prediction", ref_tm.cell_prediction, ora_tm.cell_prediction)
    eq("active_cell cols", ref_tm.active_cell[0], ora_tm.active_cell[0])
    eq("active_cell cells", ref_tm.active_cell[1], ora_tm.active_cell[1])
    eq("winner_cell cols", ref_tm.winner_cell[0], ora_tm.winner_cell[0])
    eq("winner_cell cells", ref_tm.winner_cell[1], ora_tm.winner_cell[1])
    rd, od = ref_tm.distal_state, ora_tm.distal_state
    eq("segment_potential", rd.segment_potential, od.segment_potential)
    eq("matching_segment", rd.matching_segment, od.matching_segment)
    eq("matching_segment_activation", rd.matching_segment_activation, od.matching_segment_activation)
    eq("matching_segment_active", rd.matching_segment_active, od.matching_segment_active)
    eq("prediction", rd.prediction, od.prediction)
    eq("max_jittered_potential", rd.max_jittered_potential, od.max_jittered_potential)
    eq("matching_segment_jittered_potential", rd.matching_segment_jittered_potential,
       od.matching_segment_jittered_potential)


def compare_store(t, ref_htm, ora):
    seg_cell, presyn, perm, nsyn, bundle_segments = reference_store(ref_htm.temporal_memory)
    otm = ora.temporal_memory
    S = otm.S
    assert len(seg_cell) == S, f"step {t}: segment count {len(seg_cell)} vs {S}"
    assert np.array_equal(seg_cell, otm.seg_cell[:S]), f"step {t}: seg_cell"
    assert np.array_equal(nsyn, otm.seg_nsyn[:S]), f"step {t}: seg_nsyn"
    assert np.array_equal(bundle_segments, otm.segcount), f"step {t}: segcount"
    a = canonical_synapses(seg_cell, presyn, perm)
    b = canonical_synapses(otm.seg_cell[:S], otm.presyn[:S], otm.perm[:S])
    for s, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0] and np.array_equal(x[1], y[1]), f"step {t}: segment {s} presyn"
        assert np.array_equal(x[2].view(np.int32), y[2].view(np.int32)), f"step {t}: segment {s} perm bits"
    rsp, osp = ref_htm.spatial_pooler, ora.spatial_pooler
    assert np.array_equal(rsp.proximal_projection.permanence, osp.permanence), f"step {t}: SP permanence"
    assert np.array_equal(rsp.boosting.duty_cycle.view(np.int32), osp.duty_cycle.view(np.int32)), f"step {t}: duty"


def build_pair(ref, seed, input_dim, column_dim, cell_dim, active_columns=None,
               sp_params=None, tm_params=None):
    """Reference HTM (with hooks) and oracle HTM sharing one initial SP permanence matrix.

    Non-default parameters reach the reference through its own constructors
    (DenseProjection projections.py:7-11, PredictiveProjection :205-210) and hooks
    (networks.py:16,50,134)."""
    if active_columns is None:
        active_columns = round(column_dim * 0.02)
    np.random.seed(seed)
    kw = {}
    boost_kw = {}
    if sp_params is not None:
        kw["proximal_projection"] = ref.projections.DenseProjection(
            input_dim, column_dim, permanence_mean=sp_params.permanence_mean,
            permanence_std=sp_params.permanence_std, permanence_threshold=sp_params.permanence_threshold,
            permanence_increment=sp_params.permanence_increment,
            permanence_decrement=sp_params.permanence_decrement)
        boost_kw = dict(intensity=sp_params.boost_intensity, momentum=sp_params.boost_momentum)
    sp = ref.networks.SpatialPooler(
        input_dim, column_dim, active_columns,
        boosting=DocumentedExpBoosting(column_dim, active_columns, **boost_kw),
        inhibition=StableTopK(active_columns), **kw)
    tm = None
    if tm_params is not None:
        tm = ref.networks.TemporalMemory(column_dim, cell_dim, distal_projection=ref.projections.PredictiveProjection(
            column_dim * cell_dim, **{f: getattr(tm_params, f) for f in tm_params.__dataclass_fields__}))
    ref_htm = ref.networks.HierarchicalTemporalMemory(
        input_dim, column_dim, cell_dim, active_columns=active_columns, spatial_pooler=sp, temporal_memory=tm)
    ora = HTMOracle(input_dim, column_dim, cell_dim, active_columns=active_columns, seed=seed,
                    sp_params=sp_params, tm_params=tm_params,
                    permanence=sp.proximal_projection.permanence.copy())
    return ref_htm, ora


def pattern_index(t, patterns, jump, rng):
    """Cyclic pattern order; with probability `jump` a random pattern instead (mispredictions)."""
    if jump > 0.0 and rng.rand() < jump:
        return int(rng.randint(patterns))
    return t % patterns


def run_lockstep(ref, seed, input_dim, column_dim, cell_dim, patterns, density, noise, steps,
                 store_every=10, learning_schedule=None, record=None, sp_params=None, tm_params=None,
                 jump=0.0, epsilon=None, prev_schedule=None):
    """Run both for `steps` timesteps; assert equality of every output each step.

    `record(t, x, ref_sp, ref_tm, ref_htm)` is called after every step if given.
    Returns statistics used to assert that the run contains no implementation-defined choice.
    """
    ref_htm, ora = build_pair(ref, seed, input_dim, column_dim, cell_dim, sp_params=sp_params, tm_params=tm_params)
    if epsilon is not None:                          # TemporalMemory.process(epsilon=) (networks.py:91): a Python float there
        ora.temporal_memory.eps = np.float32(epsilon)
    bank, rng = make_inputs(seed + 1, patterns, input_dim, density)
    stats = dict(ambiguous_topk=0, steps=steps)
    ref_hist, ora_hist = [ref_htm.temporal_memory.get_empty_state()], [None]      # states after step t - 1 (index t); [0] = the empty state
    with keyed_rand(seed, cell_dim) as patch:
        for t in range(steps):
            x = bank[pattern_index(t, patterns, jump, rng)] ^ (rng.rand(input_dim) < noise)
            learning = True if learning_schedule is None else bool(learning_schedule(t))
            patch.step = t
            back = None if prev_schedule is None else prev_schedule(t)     # TemporalMemory.process(prev_state=): the state of `back` steps ago
            if epsilon is None and back is None:
                ref_sp, ref_tm = ref_htm.process(x, learning=learning)
            else:                                    # (HierarchicalTemporalMemory.process passes neither on: the two layers by hand, networks.py:146-149)
                ref_sp = ref_htm.spatial_pooler.process(x, learning=learning)
                kw = {} if epsilon is None else dict(epsilon=epsilon)
                if back is not None:
                    kw["prev_state"] = ref_hist[max(len(ref_hist) - back, 1)]
                ref_tm = ref_htm.temporal_memory.process(ref_sp, learning=learning, **kw)
            if back is None:
                ora_sp, ora_tm = ora.step(x, learning=learning)
            else:
                ora_sp = ora.spatial_pooler.step(x, learning=learning)
                ora_tm = ora.temporal_memory.step(ora_sp.active_column, learning=learning, prev_state=ora_hist[max(len(ora_hist) - back, 1)])
            ref_hist.append(ref_tm)
            ora_hist.append(ora_tm)
            compare_step(t, ref_sp, ref_tm, ora_sp, ora_tm, cell_dim)
            if t % store_every == 0 or t == steps - 1:
                compare_store(t, ref_htm, ora)
            if record is not None:
                record(t, x, learning, ref_sp, ref_tm, ref_htm)
    stats["ambiguous_topk"] = ref_htm.spatial_pooler.inhibition.ambiguous_calls
    stats["segments"] = ora.temporal_memory.S
    stats["slots"] = ora.temporal_memory.slots
    return stats, ref_htm, ora
