"""Adapters that put the HIP path (through the Python drop-in classes and the C ABI) behind the
replay interface of tests/golden_replay.py, plus an oracle-vs-GPU lock-step runner.
Not collected by pytest."""

import numpy as np

import golden_replay as gr
from oracle import HTMOracle, SPParams, TMParams, canonical_synapses


def make_htm(input_dim, column_dim, cell_dim, active_columns, seed, permanence, sp_params=None, tm_params=None,
             segment_capacity=None, segment_slots=128):
    import bithtm_amd as B
    spp, tmp = sp_params or SPParams(), tm_params or TMParams()
    proximal = B.DenseProjection(input_dim, column_dim, permanence_threshold=spp.permanence_threshold,
                                 permanence_increment=spp.permanence_increment,
                                 permanence_decrement=spp.permanence_decrement)
    proximal.permanence = permanence
    sp = B.SpatialPooler(input_dim, column_dim, active_columns, proximal_projection=proximal,
                         boosting=B.ExponentialBoosting(column_dim, active_columns, intensity=spp.boost_intensity,
                                                        momentum=spp.boost_momentum))
    distal = B.PredictiveProjection(column_dim * cell_dim, segment_capacity=segment_capacity, segment_slots=segment_slots,
                                    **{f: getattr(tmp, f) for f in tmp.__dataclass_fields__})
    tm = B.TemporalMemory(column_dim, cell_dim, distal_projection=distal, seed=seed)
    return B.HierarchicalTemporalMemory(input_dim, column_dim, cell_dim, active_columns=active_columns,
                                        spatial_pooler=sp, temporal_memory=tm)


def step_outputs(sp, tm, K):
    """Normalise State objects (reference-shaped) into the replay dictionary."""
    d = tm.distal_state
    return dict(
        active_column=sp.active_column, overlaps=sp.overlaps, boosted=sp.boosted_overlaps,
        bursting=tm.active_column_bursting[:, 0],
        act_bits=np.packbits(tm.cell_activation.reshape(-1), bitorder="little"),
        pred_bits=np.packbits(tm.cell_prediction.reshape(-1), bitorder="little"),
        winner=tm.winner_cell[0] * K + tm.winner_cell[1],
        matching=d.matching_segment, match_pot=d.segment_potential[d.matching_segment],
        match_act=d.matching_segment_activation, match_active=d.matching_segment_active,
        S=len(d.segment_potential))


class HipImpl:
    def __init__(self, g, **kw):
        self.K = int(g["cell_dim"])
        self.htm = make_htm(int(g["input_dim"]), int(g["column_dim"]), self.K, int(g["active_columns"]), int(g["seed"]),
                            gr.initial_permanence(g), g["sp_params"], g["tm_params"], **kw)

    def step(self, x, learning):
        sp, tm = self.htm.process(x, learning=learning)
        return step_outputs(sp, tm, self.K)

    def store(self):
        eng = self.htm.engine
        eng.check_capacity()
        st = eng.read_store()
        d = eng.read_distal()
        st.update(duty=eng.read_duty_cycle(), sp_permanence=eng.get_permanence(),
                  segment_potential=d["segment_potential"], max_jittered_potential=d["max_jittered_potential"])
        return st


def compare_with_oracle(t, o_sp, o_tm, h_sp, h_tm, K):
    a, b = gr.OracleImpl.__dict__, None  # noqa: F841  (kept simple: compare dictionaries)
    od = o_tm.distal_state
    want = dict(
        active_column=o_sp.active_column, overlaps=o_sp.overlaps, boosted=o_sp.boosted_overlaps,
        bursting=o_tm.active_column_bursting[:, 0],
        act_bits=np.packbits(o_tm.cell_activation.reshape(-1), bitorder="little"),
        pred_bits=np.packbits(o_tm.cell_prediction.reshape(-1), bitorder="little"),
        winner=o_tm.winner_cell[0] * K + o_tm.winner_cell[1],
        matching=od.matching_segment, match_pot=od.segment_potential[od.matching_segment],
        match_act=od.matching_segment_activation, match_active=od.matching_segment_active,
        S=len(od.segment_potential))
    got = step_outputs(h_sp, h_tm, K)
    for key in gr.FIELDS:
        x, y = np.asarray(got[key]), np.asarray(want[key])
        if key == "boosted":
            x, y = x.view(np.int64), y.view(np.int64)
        assert x.shape == y.shape, f"step {t}: {key} shape {x.shape} vs {y.shape}"
        assert np.array_equal(x, y), f"step {t}: {key} differs ({int((x != y).sum())} elements)"


def compare_store_with_oracle(t, ora, htm):
    eng = htm.engine
    eng.check_capacity()
    st = eng.read_store()
    otm, osp = ora.temporal_memory, ora.spatial_pooler
    S = otm.S
    assert st["S"] == S, f"step {t}: S {st['S']} vs {S}"
    assert np.array_equal(st["seg_cell"], otm.seg_cell[:S]), f"step {t}: seg_cell"
    assert np.array_equal(st["seg_nsyn"], otm.seg_nsyn[:S]), f"step {t}: seg_nsyn"
    assert np.array_equal(st["segcount"], otm.segcount), f"step {t}: segcount"
    a = canonical_synapses(st["seg_cell"], st["presyn"], st["perm"])
    b = canonical_synapses(otm.seg_cell[:S], otm.presyn[:S], otm.perm[:S])
    for s, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x[1], y[1]), f"step {t}: segment {s} presynaptic ids"
        assert np.array_equal(x[2].view(np.int32), y[2].view(np.int32)), f"step {t}: segment {s} permanence bits"
    assert np.array_equal(eng.read_duty_cycle().view(np.int32), osp.duty_cycle.view(np.int32)), f"step {t}: duty"
    assert np.array_equal(eng.get_permanence().view(np.int64), osp.permanence.view(np.int64)), f"step {t}: SP permanence"


def lockstep_vs_oracle(seed, input_dim, column_dim, cell_dim, patterns, density, noise, steps, store_every=20,
                       jump=0.0, learning_schedule=None, sp_params=None, tm_params=None, **kw):
    """Same seeded inputs through the oracle and the HIP path; every output compared each step."""
    active_columns = round(column_dim * 0.02)
    np.random.seed(seed)
    ora = HTMOracle(input_dim, column_dim, cell_dim, active_columns=active_columns, seed=seed,
                    sp_params=sp_params, tm_params=tm_params)
    htm = make_htm(input_dim, column_dim, cell_dim, active_columns, seed, ora.spatial_pooler.permanence.copy(),
                   sp_params, tm_params, **kw)
    rng = np.random.RandomState(seed + 1)
    bank = rng.rand(patterns, input_dim) < density
    for t in range(steps):
        idx = int(rng.randint(patterns)) if (jump > 0 and rng.rand() < jump) else t % patterns
        x = bank[idx] ^ (rng.rand(input_dim) < noise)
        learning = True if learning_schedule is None else bool(learning_schedule(t))
        o_sp, o_tm = ora.step(x, learning=learning)
        h_sp, h_tm = htm.process(x, learning=learning)
        compare_with_oracle(t, o_sp, o_tm, h_sp, h_tm, cell_dim)
        if t % store_every == 0 or t == steps - 1:
            compare_store_with_oracle(t, ora, htm)
    return ora, htm
