"""GPU: the stand-alone SpatialPooler / TemporalMemory classes (htm_sp_step / htm_tm_step), the
learning / return_winner_cell switches, edge-case inputs and the loud failure modes."""

import os

import numpy as np
import pytest

from oracle import SpatialPoolerOracle, TemporalMemoryOracle, HTMOracle, SPParams, canonical_synapses

pytestmark = pytest.mark.gpu


def test_standalone_spatial_pooler_matches_oracle():
    import bithtm_amd as B
    I, C, k = 777, 3000, 60                      # neither a multiple of 32 / 256
    np.random.seed(1)
    sp = B.SpatialPooler(I, C, k)
    ora = SpatialPoolerOracle(I, C, k, permanence=sp.proximal_projection.permanence.copy())
    rng = np.random.RandomState(2)
    for t in range(120):
        x = rng.rand(I) < 0.1
        learning = t % 7 != 3
        got = sp.process(x, learning=learning)
        want = ora.step(x, learning=learning)
        assert got.active_column.dtype == np.int64 and got.overlaps.dtype == np.int64 and got.boosted_overlaps.dtype == np.float64
        assert np.array_equal(got.active_column, want.active_column), t
        assert np.array_equal(got.overlaps, want.overlaps), t
        assert np.array_equal(got.boosted_overlaps.view(np.int64), want.boosted_overlaps.view(np.int64)), t
    assert np.array_equal(sp.proximal_projection.permanence.view(np.int64), ora.permanence.view(np.int64))
    assert np.array_equal(sp.boosting.duty_cycle.view(np.int32), ora.duty_cycle.view(np.int32))


def test_select_keys_stay_exact_for_extreme_boost_factors():
    """The select works on `select_key(boosted)`, the double's bits with the exponent packed into 8 bits.  That is
    exact for every value a float32 factor times an integer overlap can take -- down to denormal factors and
    factors that underflow to zero.  Boost intensity 4 at 2 % density makes exp(-200 x duty): duty cycles written
    between 0 and 1.2 put the factors anywhere from 1 through the float32 denormals to 0."""
    import bithtm_amd as B
    I, C, k = 300, 4096, 82
    spp = SPParams(boost_intensity=4.0)
    np.random.seed(5)
    sp = B.SpatialPooler(I, C, k, boosting=B.ExponentialBoosting(C, k, intensity=spp.boost_intensity, momentum=spp.boost_momentum))
    ora = SpatialPoolerOracle(I, C, k, params=spp, permanence=sp.proximal_projection.permanence.copy())
    rng = np.random.RandomState(6)
    x = rng.rand(I) < 0.2
    sp.process(x)                                     # creates the engine
    ora.step(x)
    duty = (rng.rand(C) * 1.2).astype(np.float32)
    duty[rng.rand(C) < 0.3] = 0.0                     # plenty of factors of exactly 1 as well
    ora.duty_cycle = duty.copy()
    sp._engine.write(4, duty, np.float32)             # HTM_F_DUTY_CYCLE
    factors = set()
    for t in range(40):
        x = rng.rand(I) < 0.2
        got, want = sp.process(x), ora.step(x)
        assert np.array_equal(got.active_column, want.active_column), t
        assert np.array_equal(got.boosted_overlaps.view(np.int64), want.boosted_overlaps.view(np.int64)), t
        nz = want.boosted_overlaps[want.overlaps > 0] / want.overlaps[want.overlaps > 0]
        factors.update(np.frexp(nz[nz > 0])[1].tolist())
    assert min(factors) < -126 and max(factors) >= 0, (min(factors), max(factors))      # denormal factors were in play


def test_standalone_temporal_memory_matches_oracle_with_unsorted_columns():
    """TemporalMemory.process(sp_state) fed by a foreign SP: any object with `.active_column`."""
    import bithtm_amd as B
    from types import SimpleNamespace
    C, K, k = 1024, 16, 24
    tm = B.TemporalMemory(C, K, seed=5)
    ora = TemporalMemoryOracle(C, K, seed=5)
    rng = np.random.RandomState(3)
    seqs = [np.sort(rng.choice(C, k, replace=False)) for _ in range(12)]
    assert tm.last_state.cell_prediction.shape == (C, K) and not tm.last_state.cell_prediction.any()
    for t in range(150):
        cols = seqs[t % 12]
        learning = t % 11 != 4
        rwc = not (t % 13 == 6 and not learning)
        shuffled = cols[rng.permutation(k)]                       # reference order is arbitrary (argpartition)
        got = tm.process(SimpleNamespace(active_column=shuffled), learning=learning, return_winner_cell=rwc)
        want = ora.step(cols, learning=learning, return_winner_cell=rwc)
        assert np.array_equal(got.cell_activation, want.cell_activation), t
        assert np.array_equal(got.cell_prediction, want.cell_prediction), t
        # per-column results come back in the CALLER's column order, as with the reference (networks.py:96-97,103-104,116-117)
        back = np.searchsorted(cols, shuffled)
        assert np.array_equal(got.active_column_bursting, want.active_column_bursting[back]), t
        rows, cells = np.where(want.cell_activation[shuffled])
        assert np.array_equal(got.active_cell[0], shuffled[rows]) and np.array_equal(got.active_cell[1], cells), t
        if learning or rwc:
            wm = np.zeros((C, K), dtype=bool)
            wm[want.winner_cell] = True
            rows, cells = np.where(wm[shuffled])
            assert np.array_equal(got.winner_cell[0], shuffled[rows]) and np.array_equal(got.winner_cell[1], cells), t
        else:
            assert got.winner_cell is None and want.winner_cell is None
        d, od = got.distal_state, want.distal_state
        assert np.array_equal(d.matching_segment, od.matching_segment), t
        assert np.array_equal(d.segment_potential, od.segment_potential), t
        assert np.array_equal(d.prediction, od.prediction), t
        assert np.array_equal(d.max_jittered_potential.view(np.int32), od.max_jittered_potential.view(np.int32)), t
        assert tm.last_state is got
    st = tm._engine.read_store()
    a = canonical_synapses(st["seg_cell"], st["presyn"], st["perm"])
    b = canonical_synapses(ora.seg_cell[:ora.S], ora.presyn[:ora.S], ora.perm[:ora.S])
    assert len(a) == len(b) and all(x[0] == y[0] and np.array_equal(x[1], y[1]) and np.array_equal(x[2].view(np.int32), y[2].view(np.int32)) for x, y in zip(a, b))
    assert tm.flatten_cell((np.array([2, 3]), np.array([1, 5]))).tolist() == [2 * K + 1, 3 * K + 5]


def test_temporal_memory_epsilon_matches_oracle():
    """TemporalMemory.process(..., epsilon=) (networks.py:91): ties a third of a count wide -- several winner cells per
    bursting column, several "best" segments per cell -- then back to the default, against the oracle (which the CPU
    suite pins against the reference run with the same epsilon)."""
    import bithtm_amd as B
    from types import SimpleNamespace
    C, K, k = 1024, 8, 24
    tm = B.TemporalMemory(C, K, seed=9)
    ora = TemporalMemoryOracle(C, K, seed=9)
    rng = np.random.RandomState(10)
    seqs = [np.sort(rng.choice(C, k, replace=False)) for _ in range(10)]
    multi = 0
    for t in range(160):
        eps = 1e-8 if 100 <= t < 130 else 0.3
        cols = seqs[t % 10] if rng.rand() > 0.1 else seqs[rng.randint(10)]
        ora.eps = np.float32(eps)
        got = tm.process(SimpleNamespace(active_column=cols), epsilon=eps)
        want = ora.step(cols)
        assert np.array_equal(got.cell_activation, want.cell_activation), t
        assert np.array_equal(got.cell_prediction, want.cell_prediction), t
        wm = np.zeros((C, K), dtype=bool)
        wm[want.winner_cell] = True
        rows, cells = np.where(wm[cols])
        assert np.array_equal(got.winner_cell[0], cols[rows]) and np.array_equal(got.winner_cell[1], cells), t
        multi += int((wm[cols].sum(axis=1) > 1).sum())
        d, od = got.distal_state, want.distal_state
        assert np.array_equal(d.matching_segment, od.matching_segment), t
        assert np.array_equal(d.max_jittered_potential.view(np.int32), od.max_jittered_potential.view(np.int32)), t
    assert multi > 50                                  # (the wide ties happened)
    st = tm._engine.read_store()
    a = canonical_synapses(st["seg_cell"], st["presyn"], st["perm"])
    b = canonical_synapses(ora.seg_cell[:ora.S], ora.presyn[:ora.S], ora.perm[:ora.S])
    assert len(a) == len(b) and all(x[0] == y[0] and np.array_equal(x[1], y[1]) and np.array_equal(x[2].view(np.int32), y[2].view(np.int32)) for x, y in zip(a, b))
    with pytest.raises(NotImplementedError):
        tm.process(SimpleNamespace(active_column=cols), epsilon=2.0)


@pytest.mark.parametrize("K", [8, 40])
def test_temporal_memory_prev_state_matches_oracle(K):
    """TemporalMemory.process(sp_state, prev_state=X) (networks.py:91-93) with X an earlier State of the same object,
    or the empty state (a sequence reset): X's fields become the device's previous step.  Against the oracle, which the
    CPU suite pins against the reference run with the same prev_state schedule."""
    import bithtm_amd as B
    from types import SimpleNamespace
    C, k = 1024, 24
    tm = B.TemporalMemory(C, K, seed=11)
    ora = TemporalMemoryOracle(C, K, seed=11)
    rng = np.random.RandomState(12)
    seqs = [np.sort(rng.choice(C, k, replace=False)) for _ in range(10)]
    got_hist, want_hist = [], []
    empty = SimpleNamespace(cell_prediction=np.zeros((C, K), bool), cell_activation=np.zeros((C, K), bool), winner_cell=None, distal_state=None)
    for t in range(130):
        cols = seqs[t % 10]
        learning = t % 13 != 5
        if t in (50, 90):
            got = tm.process(SimpleNamespace(active_column=cols), prev_state=tm.get_empty_state(), learning=learning)
            want = ora.step(cols, learning=learning, prev_state=empty)
        elif t > 10 and t % 9 == 0:
            got = tm.process(SimpleNamespace(active_column=cols), prev_state=got_hist[-3], learning=learning)
            want = ora.step(cols, learning=learning, prev_state=want_hist[-3])
        else:
            got = tm.process(SimpleNamespace(active_column=cols), learning=learning)
            want = ora.step(cols, learning=learning)
        got_hist.append(got)
        want_hist.append(want)
        assert np.array_equal(got.cell_activation, want.cell_activation), t
        assert np.array_equal(got.cell_prediction, want.cell_prediction), t
        assert np.array_equal(got.active_column_bursting, want.active_column_bursting), t
        d, od = got.distal_state, want.distal_state
        assert np.array_equal(d.matching_segment, od.matching_segment), t
        assert np.array_equal(d.max_jittered_potential.view(np.int32), od.max_jittered_potential.view(np.int32)), t
    st = tm._engine.read_store()
    a = canonical_synapses(st["seg_cell"], st["presyn"], st["perm"])
    b = canonical_synapses(ora.seg_cell[:ora.S], ora.presyn[:ora.S], ora.perm[:ora.S])
    assert len(a) == len(b) and all(x[0] == y[0] and np.array_equal(x[1], y[1]) and np.array_equal(x[2].view(np.int32), y[2].view(np.int32)) for x, y in zip(a, b))


def test_sp_and_tm_objects_fuse_into_one_engine_and_compute_alias():
    import bithtm_amd as B
    np.random.seed(4)
    sp = B.SpatialPooler(100, 512, 12)
    tm = B.TemporalMemory(512, 8, seed=1)
    perm = sp.proximal_projection.permanence.copy()
    htm = B.HierarchicalTemporalMemory(100, 512, 8, active_columns=12, spatial_pooler=sp, temporal_memory=tm)
    ora = HTMOracle(100, 512, 8, active_columns=12, seed=1, permanence=perm)
    rng = np.random.RandomState(5)
    for t in range(40):
        x = rng.rand(100) < 0.2
        s, m = htm.compute(x)
        os_, om = ora.step(x)
        assert np.array_equal(s.active_column, os_.active_column) and np.array_equal(m.cell_prediction, om.cell_prediction)
    with pytest.raises(RuntimeError):
        sp.process(rng.rand(100) < 0.2)             # fused: step through the HTM object


def test_edge_inputs_all_zero_and_all_one():
    import bithtm_amd as B
    np.random.seed(6)
    htm = B.HierarchicalTemporalMemory(64, 256, 4, active_columns=8)
    ora = HTMOracle(64, 256, 4, active_columns=8, seed=0, permanence=htm.spatial_pooler.proximal_projection.permanence.copy())
    for x in (np.zeros(64, bool), np.ones(64, bool), np.zeros(64, bool), np.arange(64) % 2 == 0):
        for _ in range(3):
            s, m = htm.process(x)
            os_, om = ora.step(x)
            assert np.array_equal(s.active_column, os_.active_column)          # all-ties case: lowest indices win
            assert np.array_equal(s.overlaps, os_.overlaps)
            assert np.array_equal(m.cell_activation, om.cell_activation) and np.array_equal(m.cell_prediction, om.cell_prediction)


def test_states_stay_valid_after_later_steps_and_unread_states_cost_nothing():
    import bithtm_amd as B
    np.random.seed(7)
    htm = B.HierarchicalTemporalMemory(128, 1024, 8)
    rng = np.random.RandomState(8)
    xs = [rng.rand(128) < 0.1 for _ in range(6)]
    kept = [htm.process(x) for x in xs]                   # hold every State, read them only afterwards
    np.random.seed(7)
    htm2 = B.HierarchicalTemporalMemory(128, 1024, 8)
    for (s, m), x in zip(kept, xs):
        s2, m2 = htm2.process(x)
        assert np.array_equal(s.active_column, s2.active_column) and np.array_equal(m.cell_prediction, m2.cell_prediction)
        assert np.array_equal(m.winner_cell[0], m2.winner_cell[0])


def test_capacity_overflow_is_reported_not_truncated():
    import bithtm_amd as B
    np.random.seed(9)
    tm = B.TemporalMemory(1024, 8, distal_projection=B.PredictiveProjection(1024 * 8, segment_capacity=64, segment_slots=64))
    htm = B.HierarchicalTemporalMemory(128, 1024, 8, temporal_memory=tm)
    rng = np.random.RandomState(10)
    with pytest.raises(B.CapacityError, match="segment pool"):
        for _ in range(40):
            htm.process(rng.rand(128) < 0.1)
            htm.engine.check_capacity()


def test_bad_arguments_fail_loudly():
    import bithtm_amd as B
    tm = B.TemporalMemory(256, 4)
    from types import SimpleNamespace
    with pytest.raises(B.HtmError):
        tm.process(SimpleNamespace(active_column=np.array([1, 1, 2])))   # duplicate column
    with pytest.raises(NotImplementedError):
        tm.process(SimpleNamespace(active_column=np.array([1, 2])), epsilon=0.0)     # (0, 1] only


@pytest.mark.parametrize("eager_below", ["0", "64", "18"], ids=["graphs-for-every-call", "default-call-policy", "eager-below-18"])
def test_batched_run_equals_step_by_step(eager_below, monkeypatch):
    """htm.run (device-resident bank, hipGraph replay, the Spatial Pooler working ahead of the Temporal
    Memory) == the same inputs through process().  The run lengths exercise every launch plan: single
    steps, the two shortened steps at the end of a run, the eager cold start, single-step graphs and
    the 16-step steady-state graph with its remainders.  Under three call policies: graphs for every call that asks for them
    (what conftest.py sets for the suite), the library's default (calls of fewer than 64 steps launch eagerly whatever they
    ask: what the driver's bench command runs), and a limit in the middle of this test's run lengths."""
    import bithtm_amd as B
    monkeypatch.setenv("BITHTM_EAGER_BELOW", eager_below)
    rng = np.random.RandomState(11)
    bank = rng.rand(30, 200) < 0.08
    outs = []
    for mode in ("graph", "eager", "graph-nopipe", "mixed", "chunks", "chunks-eager", "continuing", "continuing-eager", "process",
                 "process-read", "process-plain"):
        np.random.seed(12)
        os.environ["BITHTM_DEFER_TAIL"] = "0" if mode == "process-plain" else "1"
        htm = B.HierarchicalTemporalMemory(200, 2048, 16)
        os.environ.pop("BITHTM_DEFER_TAIL")
        if mode in ("process", "process-plain"):   # a caller that steps and steps: each step's last launch rides in the next call's first
            for t in range(95):                    # ("plain": four launches per call, nothing held back)
                htm.process(bank[t % 30])
        elif mode == "process-read":               # ... and one that reads something every few steps (the held-back launch is let go)
            for t in range(95):
                sp_state, tm_state = htm.process(bank[t % 30])
                if t % 3 == 0:
                    assert tm_state.cell_prediction.shape == (2048, 16)
                if t % 7 == 0:
                    htm.engine.info()
        elif mode == "mixed":                      # batched runs, host-fed steps and another bank object in turn
            htm.run(bank, 40)
            for t in range(40, 47):
                htm.process(bank[t % 30])
            htm.run(bank, 20, use_graph=False)
            other = bank.copy()                    # a different bank object with the same content
            htm.run(other, 8)
            htm.run(bank, 20, pipeline=False)
        elif mode.startswith("chunks"):
            for n in (1, 2, 3, 1, 17, 18, 19, 34):
                htm.run(bank, n, use_graph=(mode == "chunks"))
        elif mode.startswith("continuing"):        # a caller streaming in chunks: the SP stays ahead across the calls
            g = mode == "continuing"
            for n in (1, 2, 3, 20, 1, 16, 33):
                htm.run(bank, n, use_graph=g, continuing=True)
            assert htm.temporal_memory.last_state.cell_prediction.shape == (2048, 16)     # TM fields can be read meanwhile
            with pytest.raises(B.HtmError, match="ahead"):
                htm.process(bank[0])                                                     # ... SP-dependent calls cannot
            with pytest.raises(B.HtmError, match="ahead"):
                htm.engine.read_duty_cycle()
            htm.run(bank, 1, use_graph=g)          # a one-step drain (leaves a computed front unused)
            htm.run(bank, 5, use_graph=g, continuing=True)
            htm.run(bank, 13, use_graph=g)         # 1+2+3+20+1+16+33 + 1 + 5 + 13 = 95
        else:
            htm.run(bank, 95, use_graph=mode.startswith("graph"), pipeline=(mode != "graph-nopipe"))
        st = htm.engine.read_store()
        outs.append((htm.engine.read_sp_fields()["active_column"], st["seg_cell"], st["seg_nsyn"], st["perm"],
                     htm.temporal_memory.last_state.cell_prediction, htm.engine.read_duty_cycle(),
                     htm.engine.read_sp_fields()["boosted_overlaps"]))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("env", [{}, {"BITHTM_LEAN": "0"}, {"BITHTM_SEL_WINDOW_OFFSET": "4000"}, {"BITHTM_CAND_D": "0"}, {"BITHTM_CAND_D": "1"},
                                 {"BITHTM_CAND_PAIRWISE": "0", "BITHTM_CAND_OTHERS": "0"}, {"BITHTM_CAND_PAIRWISE": "0", "BITHTM_CAND_OTHERS": "4"},
                                 {"BITHTM_CAND_PAIRWISE": "0", "BITHTM_CAND_OTHERS": "0", "BITHTM_CAND_SPECULATE": "0"},
                                 {"BITHTM_CAND_PAIRWISE": "0", "BITHTM_CAND_OTHERS": "4", "BITHTM_CAND_SPECULATE": "0"},
                                 {"BITHTM_CAND_PAIRWISE": "0", "BITHTM_CAND_ZOOM": "0"}, {"BITHTM_CAND_PAIRWISE": "2", "BITHTM_CAND_ZOOM": "3", "BITHTM_CAND_OTHERS": "0"},
                                 {"BITHTM_LEAN_SCAN": "1", "BITHTM_LEAN_LEARN": "1"}, {"BITHTM_LEAN_SCAN": "3", "BITHTM_LEAN_LEARN": "2", "BITHTM_LEAN_OVERLAP": "3"},
                                 {"BITHTM_FUSE_TM": "0"},
                                 # the streaming (large-pool) scan beside the learning role and the select finish (k_learn_scan_emit<E, 4, false>),
                                 # and in the four-launch schedule (k_scan_sel<*, 1>); ..._ABOVE: the pool outgrows the threshold in mid-run
                                 {"BITHTM_SCAN_LARGE": "1"}, {"BITHTM_SCAN_LARGE": "1", "BITHTM_LEAN": "0"}, {"BITHTM_SCAN_LARGE": "1", "BITHTM_LEAN_SCAN": "2", "BITHTM_LEAN_LEARN": "2"},
                                 {"BITHTM_SCAN_LARGE_ABOVE": "1500"}, {"BITHTM_SCAN_LARGE_ABOVE": "1500", "BITHTM_LEAN": "0"},
                                 # ... with fixed shares per scan block instead of every block joining the scan
                                 {"BITHTM_SCAN_LARGE": "1", "BITHTM_SCAN_DYN": "0"}, {"BITHTM_SCAN_LARGE": "1", "BITHTM_SCAN_DYN": "0", "BITHTM_LEAN_SCAN": "3"},
                                 # the large-pool form of the learn / punish classification (32-row words listed per block)
                                 {"BITHTM_CLASSIFY_WORDS_ABOVE": "0"}, {"BITHTM_CLASSIFY_WORDS_ABOVE": "0", "BITHTM_SCAN_LARGE": "1"},
                                 # the library's default call policy (conftest.py asks for graphs whatever the call's length)
                                 {"BITHTM_EAGER_BELOW": "64"}, {"BITHTM_EAGER_BELOW": "64", "BITHTM_LEAN": "0"},
                                 # The default is two launches per step (k_act_mid_rows: the middle role behind an in-launch fan-in of the
                                 # activation blocks, winner rows that count their own new bits against the coming input): everything above
                                 # without a BITHTM_LEAN runs it; here with grids of a few blocks and every grid order of its roles ...
                                 {"BITHTM_LEAN_OVERLAP": "3", "BITHTM_LEAN2_CLASSIFY": "2", "BITHTM_LEAN_SCAN": "3", "BITHTM_LEAN_LEARN": "2"},
                                 {"BITHTM_LEAN2_ORDER": "0123"}, {"BITHTM_LEAN2_ORDER": "3201", "BITHTM_LEAN2_CLASSIFY": "5"}, {"BITHTM_LEAN2_ORDER": "2031", "BITHTM_LEAN2_CLASSIFY": "300"},
                                 # (process() -- the side every mode is compared with -- steps through the same launch without an overlap role;
                                 # here as it was before: the activation in the select finish's blocks, the rows beside the middle role)
                                 {"BITHTM_STEP_SPLIT": "0"}, {"BITHTM_STEP_SPLIT": "0", "BITHTM_DEFER_TAIL": "0"},
                                 # ... and the three-launch schedule (k_act_rows, k_mid_overlap): alone, with a window that always misses, with
                                 # grids of a few blocks, with both forms of the classification and of the scan, under the default call policy
                                 {"BITHTM_LEAN": "1"}, {"BITHTM_LEAN": "1", "BITHTM_SEL_WINDOW_OFFSET": "4000"}, {"BITHTM_LEAN": "1", "BITHTM_EAGER_BELOW": "64"},
                                 {"BITHTM_LEAN": "1", "BITHTM_LEAN_OVERLAP": "3", "BITHTM_LEAN_SCAN": "3", "BITHTM_LEAN_LEARN": "2"},
                                 {"BITHTM_LEAN": "1", "BITHTM_CLASSIFY_WORDS_ABOVE": "0", "BITHTM_SCAN_LARGE": "1"}, {"BITHTM_LEAN": "1", "BITHTM_SCAN_LARGE_ABOVE": "1500"}],
                         ids=lambda e: ",".join(f"{k[7:]}={v}" for k, v in e.items()) or "default")
def test_pipelined_schedules_and_select_paths_equal_step_by_step(env, monkeypatch):
    """htm.run in its pipelined schedules -- two launches per step (the default), three (the learning role scanning its own rows
    beside the scan, the one-pass windowed select; both share their last launch) and the four-launch one -- against process(), with the select forced down every
    path: a window that always misses (exact fallback each step), records that overflow (slots 0 / 1), the tie merge
    (pairwise 0; others 0 = radix refinement only; SPECULATE 0 = without the shortcut "the k-th key is the heaviest key"), the
    cut to the k-th key's sub-bin that a crowded bin gets before it is ranked (ZOOM 0: every merge; 3: some), and grids of a
    few blocks (every wave loops)."""
    import bithtm_amd as B
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.RandomState(21)
    bank = rng.rand(30, 300) < 0.06
    outs = []
    for mode in ("graph", "eager", "continuing", "process"):
        np.random.seed(22)
        htm = B.HierarchicalTemporalMemory(300, 4096, 8)
        if mode == "process":
            for t in range(120):
                htm.process(bank[t % 30])
        elif mode == "continuing":
            for n in (1, 2, 40, 17, 60):
                htm.run(bank, n, continuing=n != 60)
        else:
            htm.run(bank, 120, use_graph=mode == "graph")
        info = htm.engine.check_capacity()
        if mode == "graph" and "BITHTM_SEL_WINDOW_OFFSET" in env:
            assert info.select_fallbacks >= 100          # (the knob did what it is for)
        if mode == "graph" and "BITHTM_CAND_ZOOM" in env:
            assert info.select_zoom_steps >= (100 if env["BITHTM_CAND_ZOOM"] == "0" else 10), info.select_zoom_steps
        st = htm.engine.read_store()
        d = htm.engine.read_distal()
        outs.append((htm.engine.read_sp_fields()["active_column"], st["seg_cell"], st["seg_nsyn"], st["presyn"], st["perm"],
                     htm.temporal_memory.last_state.cell_prediction, htm.engine.read_duty_cycle(), d["matching_segment"],
                     d["max_jittered_potential"], d["segment_potential"]))
    for o in outs[:-1]:
        for a, b in zip(outs[-1], o):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("lean, per_step", [(None, 2), ("1", 3), ("0", 4)], ids=["default", "LEAN=1", "LEAN=0"])
def test_launches_per_timestep_of_the_batched_run(lean, per_step, monkeypatch):
    """htm.run issues two launches per steady-state timestep by default (k_act_mid_rows + k_learn_scan_emit), three under
    BITHTM_LEAN=1, four under BITHTM_LEAN=0 -- counted from the library's own per-launch profile of an eager call (the first
    step of a call selects its winners on its own, the last steps look ahead less: the count is taken over the steps between)."""
    import bithtm_amd as B
    if lean is not None:
        monkeypatch.setenv("BITHTM_LEAN", lean)
    np.random.seed(3)
    htm = B.HierarchicalTemporalMemory(300, 4096, 8)
    bank = np.random.RandomState(4).rand(20, 300) < 0.06
    htm.run(bank, 10)
    counts = []
    for n in (10, 30):
        htm.engine.profile(True)
        htm.run(bank, n)
        counts.append(sum(cnt for _, cnt in htm.engine.profile_read().values()))
        htm.engine.profile(False)
    assert counts[1] - counts[0] == 20 * per_step, counts          # 20 more steady-state steps
    names = set()
    htm.engine.profile(True)
    htm.run(bank, 5)
    names = {n for n, (_, cnt) in htm.engine.profile_read().items() if cnt}
    htm.engine.profile(False)
    assert ("tm_activate+tm_mid+sp_learn+sp_overlap" in names) == (per_step == 2), names
    assert ("tm_mid+sp_overlap" in names) == (per_step == 3) and ("tm_scan+sp_select" in names) == (per_step == 4), names


@pytest.mark.parametrize("env", [{}, {"BITHTM_SCAN_LARGE": "1"}, {"BITHTM_SCAN_LARGE_ABOVE": "600"}, {"BITHTM_EAGER_BELOW": "64"}, {"BITHTM_LEAN": "1"},
                                 {"BITHTM_LEAN": "1", "BITHTM_EAGER_BELOW": "64"}],
                         ids=lambda e: ",".join(f"{k[7:]}={v}" for k, v in e.items()) or "default")
def test_random_call_patterns_equal_step_by_step(env, monkeypatch):
    """Seeded random sequences of htm.run (graph / eager, continuing or not, learning on or off, run lengths on either
    side of the 16-step graphs), host-fed process() steps in between, on random small shapes -- against a twin that only
    ever calls process().  (A longer sweep: BITHTM_CALL_FUZZ_SEQUENCES.)  Also with the large-pool form of the scan in every
    launch that holds one, and with a threshold the pools cross somewhere inside the sequence (the twin is created under the
    default policy)."""
    import bithtm_amd as B
    for key, v in env.items():
        monkeypatch.setenv(key, v)

    def digest(htm):
        st, d = htm.engine.read_store(), htm.engine.read_distal()
        w = max(int(st["seg_nsyn"].max(initial=0)), 1)      # (rows are packed; a default-sized pool may have grown its rows)
        assert (st["presyn"][:, w:] == -1).all()
        return (htm.engine.read_sp_fields()["active_column"], st["seg_cell"], st["seg_nsyn"], st["presyn"][:, :w], st["perm"][:, :w],
                htm.temporal_memory.last_state.cell_prediction, htm.engine.read_duty_cycle(), d["matching_segment"], d["max_jittered_potential"])

    for seed in range(int(os.environ.get("BITHTM_CALL_FUZZ_SEQUENCES", "12"))):
        rng = np.random.RandomState(1000 + seed)
        I, C, K = int(rng.choice([64, 200, 300])), int(rng.choice([1024, 2048, 4096])), int(rng.choice([4, 8, 16, 32]))
        bank = rng.rand(int(rng.choice([5, 17, 30])), I) < rng.choice([0.05, 0.1])
        np.random.seed(seed)
        a = B.HierarchicalTemporalMemory(I, C, K)
        np.random.seed(seed)
        with monkeypatch.context() as m:               # (the knobs are read when a handle is created)
            for key in env:
                if key != "BITHTM_EAGER_BELOW":
                    m.delenv(key)
            b = B.HierarchicalTemporalMemory(I, C, K)
        t, ahead = 0, False
        for _ in range(int(rng.randint(4, 12))):
            kind = "run" if ahead else str(rng.choice(["run", "run", "run", "process", "nolearn"]))
            if kind == "process":
                n = int(rng.randint(1, 6))
                for i in range(n):
                    a.process(bank[(t + i) % len(bank)])
            else:
                n = int(rng.choice([1, 2, 3, 5, 16, 17, 18, 33, 40]))
                cont, g = bool(rng.rand() < 0.5) and kind == "run", bool(rng.rand() < 0.7)
                try:
                    a.run(bank, n, learning=kind != "nolearn", use_graph=g, continuing=cont)
                except B.HtmError as e:            # (a shape the pipelined schedule does not take: no streaming either)
                    assert "needs the pipelined schedule" in str(e)
                    cont = False
                    a.run(bank, n, learning=kind != "nolearn", use_graph=g)
                ahead = cont
            for i in range(n):
                b.process(bank[(t + i) % len(bank)], learning=kind != "nolearn")
            t += n
        if ahead:
            a.run(bank, 1)
            b.process(bank[t % len(bank)])
        for x, y in zip(digest(a), digest(b)):
            assert np.array_equal(x, y), (seed, I, C, K, t)


def test_prepare_builds_the_graphs_and_runs_nothing():
    """htm_prepare (include/bithtm_hip.h): the graphs of a coming htm_run call are captured and instantiated ahead of
    time; the state does not move, and the run that follows gives what a run without it gives -- for one-shot calls and
    for a stream of continuing calls (every launch plan: cold start, steady-state graphs, the last step)."""
    import bithtm_amd as B
    rng = np.random.RandomState(41)
    bank = rng.rand(20, 200) < 0.08
    outs = []
    for prepared in (False, True):
        np.random.seed(42)
        htm = B.HierarchicalTemporalMemory(200, 2048, 16)
        eng = htm.engine
        dev = eng.upload_bank(bank)
        for n, cont in ((37, False), (3, True), (18, True), (1, True), (40, False)):
            if prepared:
                before = eng.info().step_index
                eng.prepare(dev, bank.shape[0], n, continuing=cont)
                assert eng.info().step_index == before
            eng.run(dev, bank.shape[0], n, continuing=cont)
        eng.check_capacity()
        st = eng.read_store()
        outs.append((eng.read_sp_fields()["active_column"], st["seg_cell"], st["seg_nsyn"], st["presyn"], st["perm"], eng.read_duty_cycle()))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_batched_run_with_learning_switched_off_and_on():
    """The pipelined schedule with learning=False (no SP rows, no classification, no learning launches
    doing work) between learning runs == the same schedule of flags through process()."""
    import bithtm_amd as B
    rng = np.random.RandomState(13)
    bank = rng.rand(25, 160) < 0.1
    plan = ((22, True), (1, False), (19, False), (2, True), (40, True), (18, False), (21, True))
    outs = []
    for mode in ("graph", "eager", "process"):
        np.random.seed(14)
        htm = B.HierarchicalTemporalMemory(160, 2048, 8)
        t = 0
        for n, learning in plan:
            if mode == "process":
                for _ in range(n):
                    htm.process(bank[t % 25], learning=learning)
                    t += 1
            else:
                htm.run(bank, n, learning=learning, use_graph=(mode == "graph"))
                t += n
        st = htm.engine.read_store()
        outs.append((htm.engine.read_sp_fields()["active_column"], st["seg_cell"], st["seg_nsyn"], st["perm"], st["presyn"],
                     htm.temporal_memory.last_state.cell_prediction, htm.engine.read_duty_cycle(), htm.engine.get_permanence()))
    for o in outs[:-1]:
        for a, b in zip(outs[-1], o):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("digits,slots,pairwise,others,spec,zoom", [("2", "8", "160", "128", "1", "-1"), ("3", "8", "160", "128", "1", "-1"), ("2", "0", "160", "128", "1", "-1"),
                                                                    ("2", "1", "160", "128", "1", "-1"), ("2", "8", "0", "128", "1", "-1"), ("2", "8", "0", "4", "1", "-1"), ("2", "8", "0", "0", "1", "-1"),
                                                                    ("2", "8", "0", "128", "0", "-1"), ("2", "8", "0", "4", "0", "-1"), ("2", "8", "0", "0", "0", "-1"),
                                                                    ("2", "8", "0", "128", "1", "0"), ("3", "8", "0", "0", "0", "0"), ("2", "8", "1", "0", "1", "2"), ("2", "8", "2", "128", "0", "4")])
def test_select_is_exact_on_the_record_path_and_on_the_fallback(digits, slots, pairwise, others, spec, zoom, monkeypatch):
    """k_sp_emit finishes the top-k select from per-block bucket records.  slots=0 makes every block
    with a bucket key overflow its record, so the exact in-kernel fallback runs every step; slots=1
    mixes both paths; 3 launched digits is the variant with smaller buckets; pairwise=0 merges the
    records on every step the way a many-way tie is merged (the copies of one key folded into one
    entry, then all pairs if fewer than `others` other entries remain, else radix refinement:
    others=4 mixes both, others=0 is radix only; spec=0: without the shortcut that tries the heaviest key first);
    zoom >= 0: a merge of more pairs than that is first cut to the sub-bin of the k-th key (what a crowded bin of the
    full-size model gets: more pairs than blocks by a quarter)."""
    import bithtm_amd as B
    from oracle import HTMOracle
    monkeypatch.setenv("BITHTM_SEL_LAUNCH_DIGITS", digits)
    monkeypatch.setenv("BITHTM_CAND_ZOOM", zoom)
    monkeypatch.setenv("BITHTM_CAND_D", slots)
    monkeypatch.setenv("BITHTM_CAND_PAIRWISE", pairwise)
    monkeypatch.setenv("BITHTM_CAND_OTHERS", others)
    monkeypatch.setenv("BITHTM_CAND_SPECULATE", spec)
    np.random.seed(31)
    htm = B.HierarchicalTemporalMemory(300, 4096, 8)
    ora = HTMOracle(300, 4096, 8, seed=0, permanence=htm.spatial_pooler.proximal_projection.permanence.copy())
    rng = np.random.RandomState(32)
    bank = rng.rand(40, 300) < 0.05
    for t in range(130):
        x = bank[t % 40] ^ (rng.rand(300) < 0.01)
        s, m = htm.process(x)
        os_, om = ora.step(x)
        assert np.array_equal(s.active_column, os_.active_column), (digits, slots, pairwise, others, t)
        assert np.array_equal(m.cell_prediction, om.cell_prediction), (digits, slots, pairwise, others, t)
    info = htm.engine.check_capacity()
    print(f"digits {digits} slots {slots} pairwise {pairwise} others {others} spec {spec} zoom {zoom}: fallbacks {info.select_fallbacks}, sub-bin cuts {info.select_zoom_steps}")
    assert zoom == "-1" or info.select_zoom_steps >= 10, info.select_zoom_steps


def test_example_harness_prints_the_reference_report_format():
    """bithtm_amd.example: the reference's flags and its per-step line, character for character."""
    import io
    import re
    from bithtm_amd import example
    np.random.seed(3)
    out = io.StringIO()
    example.main(["--epochs", "2", "--input_patterns", "12", "--input_dim", "200", "--column_dim", "1024", "--cell_dim", "8"], out=out)
    lines = out.getvalue().strip().splitlines()
    assert len(lines) == 2 * 12 + 1 and lines[-1].endswith(" seconds.")
    pat = re.compile(r"^epoch [ \d]\d*, pattern [ \d]\d: bursting columns: [ \d]\d, correct columns: [ \d]\d, incorrect columns: [ \d]{3}\d$")
    assert all(pat.match(l) for l in lines[:-1]), lines[0]
    assert lines[0] == "epoch 0, pattern  0: bursting columns: 20, correct columns:  0, incorrect columns:    0"
    out = io.StringIO()
    example.main(["--epochs", "2", "--input_patterns", "12", "--input_dim", "200", "--column_dim", "1024", "--cell_dim", "8", "--batched"], out=out)
    assert "timesteps/s" in out.getvalue()


def test_example_harness_swaps_in_the_users_reference_temporal_memory(monkeypatch):
    """--use_reference_implementation (example.py:30,36-37): the Temporal Memory of the user's own `bithtm.reference_implementations`
    in the `temporal_memory=` slot, the Spatial Pooler on the device.  The reference does not travel to the GPU box: a stand-in
    module with the textbook class's surface (constructor, process, last_state) takes its place here, backed by the oracle --
    so the report must equal the device's own, line for line."""
    import io
    import sys
    import types
    from bithtm_amd import example
    from oracle import TemporalMemoryOracle

    class TextbookTM:
        def __init__(self, column_dim, cell_dim):
            self.o = TemporalMemoryOracle(column_dim, cell_dim, seed=0)
            self.last_state = types.SimpleNamespace(cell_prediction=np.zeros((column_dim, cell_dim), dtype=bool))

        def process(self, sp_state, prev_state=None, learning=True):
            self.last_state = self.o.step(np.sort(np.asarray(sp_state.active_column)), learning=learning)
            return self.last_state

    pkg, mod = types.ModuleType("bithtm"), types.ModuleType("bithtm.reference_implementations")
    mod.TemporalMemory = TextbookTM
    pkg.reference_implementations = mod
    monkeypatch.setitem(sys.modules, "bithtm", pkg)
    monkeypatch.setitem(sys.modules, "bithtm.reference_implementations", mod)
    args = ["--epochs", "6", "--input_patterns", "12", "--input_dim", "200", "--column_dim", "1024", "--cell_dim", "8"]
    outs = []
    for extra in ([], ["--use_reference_implementation"]):
        np.random.seed(3)
        out = io.StringIO()
        example.main(args + extra, out=out)
        outs.append(out.getvalue().strip().splitlines()[:-1])
    assert len(outs[0]) == 6 * 12 and outs[0] == outs[1]
    monkeypatch.delitem(sys.modules, "bithtm.reference_implementations")
    monkeypatch.delitem(sys.modules, "bithtm")
    with pytest.raises(SystemExit, match="PYTHONPATH"):
        example.main(args + ["--use_reference_implementation"], out=io.StringIO())


@pytest.mark.parametrize("K", [8, 40])
def test_checkpoint_and_resume(tmp_path, K):
    """save() after 90 steps, then a fresh object load()s and continues exactly like the original (one and two cell words
    per column)."""
    import bithtm_amd as B
    rng = np.random.RandomState(41)
    bank = rng.rand(30, 160) < 0.1
    np.random.seed(42)
    a = B.HierarchicalTemporalMemory(160, 2048, K, seed=9)
    a.run(bank, 90)
    a.save(tmp_path / "ckpt.npz")
    np.random.seed(43)                                   # different initial permanences on purpose
    b = B.HierarchicalTemporalMemory(160, 2048, K, seed=9)
    b.load(tmp_path / "ckpt.npz")
    for t in range(90, 150):
        sa, ma = a.process(bank[t % 30])
        sb, mb = b.process(bank[t % 30])
        assert np.array_equal(sa.active_column, sb.active_column), t
        assert np.array_equal(ma.cell_prediction, mb.cell_prediction), t
        assert np.array_equal(ma.winner_cell[0], mb.winner_cell[0]) and np.array_equal(ma.winner_cell[1], mb.winner_cell[1]), t
    da, db = a.state_dict(), b.state_dict()
    assert da.keys() == db.keys()
    for k in da:
        assert np.array_equal(da[k], db[k]), k


def test_random_small_configurations_agree_with_the_oracle():
    """A seeded sample of tests/fuzz_parity.py (cell_dim 1..64, odd input sizes, low thresholds, 64..256
    slots): process() in lock-step with the oracle, then batched pipelined runs of odd lengths."""
    import fuzz_parity
    from bithtm_amd.engine import CapacityError
    rng = np.random.RandomState(2024)
    done = 0
    for _ in range(20):
        cfg = fuzz_parity.draw_config(rng)
        seed = int(rng.randint(1 << 20))
        try:
            fuzz_parity.run_one(cfg, seed)
            done += 1
        except CapacityError:                      # a configuration that outgrows its fixed pool: documented, not parity
            pass
    assert done >= 15


def test_prev_state_adoption_keeps_the_sticky_capacity_flags():
    """TemporalMemory.process(prev_state=X) writes X back through the state import; that must not forgive an overflow of
    the segment pool the handle keeps (the import of a checkpoint, which replaces the store, does)."""
    import bithtm_amd as B
    from types import SimpleNamespace
    C, K, k = 512, 8, 20                                # (20 winners: a new segment grows 20 synapses, past the matching threshold, so it is not recycled)
    tm = B.TemporalMemory(C, K, distal_projection=B.PredictiveProjection(C * K, segment_capacity=64, segment_slots=64), seed=4)
    rng = np.random.RandomState(5)
    seqs = [np.sort(rng.choice(C, k, replace=False)) for _ in range(30)]
    first = tm.process(SimpleNamespace(active_column=seqs[0]))
    first.cell_prediction                               # (materialised: an earlier State of this object)
    for cols in seqs[1:]:
        tm.process(SimpleNamespace(active_column=cols))  # every column bursts: 20 new segments per step, the pool holds 64
    with pytest.raises(B.CapacityError, match="segment pool"):
        tm._engine.check_capacity()
    st = tm.process(SimpleNamespace(active_column=seqs[3]), prev_state=first)
    assert tm._engine.info().capacity_error & 1
    with pytest.raises(B.CapacityError, match="segment pool"):
        st.cell_prediction


def test_winner_cells_survive_a_checkpoint_round_trip(tmp_path):
    """State.winner_cell is read from the winner words; a state import fills them from the imported winner list."""
    import bithtm_amd as B
    rng = np.random.RandomState(21)
    bank = rng.rand(12, 160) < 0.1
    np.random.seed(22)
    a = B.HierarchicalTemporalMemory(160, 1024, 8)
    for t in range(40):
        _, tm_state = a.process(bank[t % 12])
    want = tm_state.winner_cell
    a.save(str(tmp_path / "ckpt.npz"))
    np.random.seed(23)
    b = B.HierarchicalTemporalMemory(160, 1024, 8)
    b.load(str(tmp_path / "ckpt.npz"))
    got = b.temporal_memory.last_state.winner_cell
    assert got is not None and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    a.load(str(tmp_path / "ckpt.npz"))                  # into the handle that wrote it, too
    got = a.temporal_memory.last_state.winner_cell
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    sa, ta = a.process(bank[4])
    sb, tb = b.process(bank[4])
    assert np.array_equal(ta.winner_cell[0], tb.winner_cell[0]) and np.array_equal(ta.cell_prediction, tb.cell_prediction)


def test_a_stream_of_continuing_runs_survives_another_handle_appearing():
    """A handle whose Spatial Pooler is ahead (HTM_RUN_CONTINUE) when another handle with its own stream is created on the
    device loses the in-kernel select finish: the next htm_run must finish the step that was begun and go on unpipelined,
    with the same results -- not return HTM_ERR_STATE for ever."""
    import bithtm_amd as B
    rng = np.random.RandomState(31)
    bank = rng.rand(20, 200) < 0.08
    np.random.seed(32)
    twin = B.HierarchicalTemporalMemory(200, 2048, 16)
    for t in range(57):
        twin.process(bank[t % 20])
    want = twin.engine.export_tm_state()
    del twin
    np.random.seed(32)
    htm = B.HierarchicalTemporalMemory(200, 2048, 16)
    htm.run(bank, 20, continuing=True)
    np.random.seed(33)
    other = B.HierarchicalTemporalMemory(200, 2048, 16)  # a second live handle, private stream
    other.process(bank[0])
    htm.run(bank, 17, continuing=True)                   # (the promise is kept; the schedule is gone)
    htm.run(bank, 20)
    got = htm.engine.export_tm_state()
    for key in want:
        assert np.array_equal(np.asarray(got[key]), np.asarray(want[key])), key
    del other


def test_default_sized_pools_grow_like_the_references_arrays():
    """utils.py:113-135: the reference's segment store grows on demand.  A pool whose size the user did not fix follows it
    (the engine is re-created with a larger pool and the state handed over, between host-fed steps and between the batches
    of a run()): more segments than the first pool held, more synapses than its rows held, the oracle's result throughout."""
    from hip_impl import compare_store_with_oracle, make_htm
    from oracle import HTMOracle, TMParams
    I, C, K, seed = 96, 1024, 4, 41
    k = round(C * 0.02)
    tmp = TMParams(segment_activation_threshold=3, segment_matching_threshold=3, segment_sampling_synapses=32, permanence_decrement=0.0)
    np.random.seed(seed)
    ora = HTMOracle(I, C, K, active_columns=k, seed=seed, tm_params=tmp)
    htm = make_htm(I, C, K, k, seed, ora.spatial_pooler.permanence.copy(), None, tmp, segment_capacity=None, segment_slots=128)
    first = htm.engine
    assert first.segment_capacity == 10240 and first.segment_slots == 128
    rng = np.random.RandomState(seed + 1)
    t = 0
    for rnd in range(2):                                # novel input throughout: every column bursts, 20 new segments per step
        bank = rng.rand(300, I) < 0.15
        for _ in range(150):                            # host-fed steps
            ora.step(bank[t % 300])
            htm.process(bank[t % 300])
            t += 1
        for _ in range(150):                            # and a device-side batch
            ora.step(bank[t % 300])
            t += 1
        htm.run(bank, 150)
        compare_store_with_oracle(t, ora, htm)
    assert htm.engine is not first and htm.engine.segment_capacity > 10240 and ora.temporal_memory.S > 10240
    small = rng.rand(6, I) < 0.15                       # a few patterns in random order, nothing ever pruned: the segments fill up
    for _ in range(400):
        x = small[rng.randint(6)]
        ora.step(x)
        htm.process(x)
        t += 1
    compare_store_with_oracle(t, ora, htm)
    assert htm.engine.segment_slots > 128 and ora.temporal_memory.seg_nsyn.max() > 128
    # an explicit capacity stays a hard limit
    import bithtm_amd as B
    fixed = B.HierarchicalTemporalMemory(I, C, K, temporal_memory=B.TemporalMemory(C, K, distal_projection=B.PredictiveProjection(
        C * K, segment_capacity=512, segment_activation_threshold=3, segment_matching_threshold=3)))
    bank = rng.rand(200, I) < 0.15
    with pytest.raises(B.CapacityError):
        for i in range(200):
            fixed.process(bank[i])
            if i % 20 == 0:
                fixed.engine.check_capacity()


@pytest.mark.parametrize("K", [33, 48, 64])
def test_up_to_64_cells_per_column_run_the_fused_step(K):
    """networks.py:53 has no cap on cell_dim.  Up to 64 cells per column the device's fused step takes the model as it is: two
    32-bit words of cells per column, a wave per active column (a half-wave up to 32).  Through process(), through run() in
    its pipelined schedules (graph / eager, continuing calls, the four-launch schedule, the streaming scan) and as a
    stand-alone Temporal Memory, against the oracle."""
    import bithtm_amd as B
    from types import SimpleNamespace
    from hip_impl import compare_store_with_oracle, compare_with_oracle
    from oracle import HTMOracle, TMParams, TemporalMemoryOracle, canonical_synapses
    I, C, seed = 200, 2048, 40 + K
    k = round(C * 0.02)
    tmp = TMParams(segment_activation_threshold=9, segment_matching_threshold=7, segment_sampling_synapses=18, permanence_punishment=0.1)
    kw = {f: getattr(tmp, f) for f in tmp.__dataclass_fields__}
    np.random.seed(seed)
    perm = np.random.randn(C, I) * 0.1
    rng = np.random.RandomState(seed + 1)
    bank = rng.rand(12, I) < 0.08

    def make():
        prox = B.DenseProjection(I, C)
        prox.permanence = perm
        tm = B.TemporalMemory(C, K, distal_projection=B.PredictiveProjection(C * K, segment_slots=64, **kw), seed=seed)
        assert tm._own_distal
        return B.HierarchicalTemporalMemory(I, C, K, active_columns=k, spatial_pooler=B.SpatialPooler(I, C, k, proximal_projection=prox), temporal_memory=tm)

    # process(): every output of every step
    ora = HTMOracle(I, C, K, active_columns=k, seed=seed, permanence=perm.copy(), tm_params=tmp)
    htm = make()
    assert htm.engine is not None
    for t in range(100):
        x = bank[t % 12] ^ (rng.rand(I) < 0.01)
        learning = t % 17 != 5
        o_sp, o_tm = ora.step(x, learning=learning)
        h_sp, h_tm = htm.process(x, learning=learning)
        compare_with_oracle(t, o_sp, o_tm, h_sp, h_tm, K)
        if t % 33 == 0 or t == 99:
            compare_store_with_oracle(t, ora, htm)
    assert ora.temporal_memory.S > 200
    # run(): the batched schedules against process()
    import os
    outs = []
    for mode, env in (("process", {}), ("graph", {}), ("eager", {}), ("continuing", {}), ("graph", {"BITHTM_LEAN": "0"}), ("graph", {"BITHTM_SCAN_LARGE": "1"}),
                      ("graph", {"BITHTM_SCAN_LARGE": "1", "BITHTM_SCAN_DYN": "0"}), ("graph", {"BITHTM_FUSE_TM": "0"}), ("graph", {"BITHTM_LEAN": "1"}),
                      ("eager", {"BITHTM_LEAN": "1"})):
        os.environ.update(env)
        try:
            htm = make()
        finally:
            for key in env:
                os.environ.pop(key)
        if mode == "process":
            for t in range(75):
                htm.process(bank[t % 12])
        elif mode == "continuing":
            for n in (1, 2, 30, 17, 25):
                htm.run(bank, n, continuing=n != 25)
        else:
            htm.run(bank, 75, use_graph=mode == "graph")
        htm.engine.check_capacity()
        st, d = htm.engine.read_store(), htm.engine.read_distal()
        outs.append((htm.engine.read_sp_fields()["active_column"], st["seg_cell"], st["seg_nsyn"], st["presyn"], st["perm"],
                     htm.temporal_memory.last_state.cell_prediction, htm.temporal_memory.last_state.cell_activation, htm.engine.read_duty_cycle(),
                     d["matching_segment"], d["max_jittered_potential"], d["segment_potential"]))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert np.array_equal(a, b)
    # stand-alone Temporal Memory (htm_tm_step), learning switched off now and then
    tm = B.TemporalMemory(C, K, distal_projection=B.PredictiveProjection(C * K, segment_slots=64, **kw), seed=seed)
    ota = TemporalMemoryOracle(C, K, tmp, seed=seed)
    seqs = [np.sort(rng.choice(C, k, replace=False)) for _ in range(7)]
    for t in range(70):
        cols = seqs[int(rng.randint(7))] if rng.rand() < 0.1 else seqs[t % 7]
        want = ota.step(cols, learning=t % 11 != 4)
        got = tm.process(SimpleNamespace(active_column=cols), learning=t % 11 != 4)
        assert got.cell_prediction.shape == (C, K)
        assert np.array_equal(got.cell_prediction, want.cell_prediction) and np.array_equal(got.cell_activation, want.cell_activation), t
        assert np.array_equal(got.winner_cell[0] * K + got.winner_cell[1], want.winner_cell[0] * K + want.winner_cell[1]), t
        assert np.array_equal(got.distal_state.matching_segment, want.distal_state.matching_segment), t
        assert np.array_equal(np.asarray(got.distal_state.max_jittered_potential).view(np.int32), want.distal_state.max_jittered_potential.view(np.int32)), t
    st = tm._engine.read_store()
    S = ota.S
    assert st["S"] == S > 50 and np.array_equal(st["seg_cell"], ota.seg_cell[:S]) and np.array_equal(st["seg_nsyn"], ota.seg_nsyn[:S])
    a, b = canonical_synapses(st["seg_cell"], st["presyn"], st["perm"]), canonical_synapses(ota.seg_cell[:S], ota.presyn[:S], ota.perm[:S])
    assert all(np.array_equal(x[1], y[1]) and np.array_equal(x[2].view(np.int32), y[2].view(np.int32)) for x, y in zip(a, b))


@pytest.mark.parametrize("K", [65, 96])
def test_more_than_64_cells_per_column(K):
    """networks.py:53 has no cap on cell_dim.  The device's fused Temporal Memory step is built on one or two 32-bit words of
    cells per column; beyond that the segment store stays on the device (it lives in cell space: htm_tm_update / htm_tm_scan on
    an engine laid out as words of 32 cells, flat cell ids unchanged) and the per-column part of TemporalMemory.process runs on
    the host.  Stand-alone and inside a HierarchicalTemporalMemory, against the oracle."""
    import bithtm_amd as B
    from types import SimpleNamespace
    from oracle import HTMOracle, TMParams, TemporalMemoryOracle, canonical_synapses
    C, k, seed = 320, 10, 17                               # (320 x 65 cells: not a whole number of 32-cell words)
    tmp = TMParams(segment_activation_threshold=6, segment_matching_threshold=5, segment_sampling_synapses=12, permanence_punishment=0.1)
    tm = B.TemporalMemory(C, K, distal_projection=B.PredictiveProjection(C * K, segment_slots=64, **{f: getattr(tmp, f) for f in tmp.__dataclass_fields__}), seed=seed)
    assert not tm._own_distal
    ora = TemporalMemoryOracle(C, K, tmp, seed=seed)
    rng = np.random.RandomState(seed)
    seqs = [np.sort(rng.choice(C, k, replace=False)) for _ in range(7)]
    for t in range(70):
        cols = seqs[int(rng.randint(7))] if rng.rand() < 0.1 else seqs[t % 7]
        want = ora.step(cols, learning=t % 11 != 4)
        got = tm.process(SimpleNamespace(active_column=cols), learning=t % 11 != 4)
        assert got.cell_prediction.shape == (C, K)
        assert np.array_equal(got.cell_prediction, want.cell_prediction) and np.array_equal(got.cell_activation, want.cell_activation), t
        assert np.array_equal(got.winner_cell[0] * K + got.winner_cell[1], want.winner_cell[0] * K + want.winner_cell[1]), t
        assert np.array_equal(got.distal_state.matching_segment, want.distal_state.matching_segment), t
        assert np.array_equal(got.distal_state.segment_potential, want.distal_state.segment_potential), t
        assert np.array_equal(np.asarray(got.distal_state.max_jittered_potential).view(np.int32), want.distal_state.max_jittered_potential.view(np.int32)), t
    st = tm.distal_projection._engine.read_store()
    S = ora.S
    assert st["S"] == S > 50 and np.array_equal(st["seg_cell"], ora.seg_cell[:S]) and np.array_equal(st["seg_nsyn"], ora.seg_nsyn[:S])
    a, b = canonical_synapses(st["seg_cell"], st["presyn"], st["perm"]), canonical_synapses(ora.seg_cell[:S], ora.presyn[:S], ora.perm[:S])
    assert all(np.array_equal(x[1], y[1]) and np.array_equal(x[2].view(np.int32), y[2].view(np.int32)) for x, y in zip(a, b))
    assert np.array_equal(tm.distal_projection.bundle_segments, ora.segcount)
    # and as the Temporal Memory of a HierarchicalTemporalMemory (the Spatial Pooler steps an engine of its own)
    I = 128
    np.random.seed(seed)
    full = HTMOracle(I, C, K, active_columns=k, seed=seed)
    prox = B.DenseProjection(I, C)
    prox.permanence = full.spatial_pooler.permanence.copy()
    htm = B.HierarchicalTemporalMemory(I, C, K, active_columns=k, spatial_pooler=B.SpatialPooler(I, C, k, proximal_projection=prox), seed=seed)
    assert htm.engine is None
    bank = rng.rand(6, I) < 0.15
    for t in range(40):
        o_sp, o_tm = full.step(bank[t % 6])
        s, m = htm.process(bank[t % 6])
        assert np.array_equal(s.active_column, o_sp.active_column) and np.array_equal(m.cell_prediction, o_tm.cell_prediction), t
        assert np.array_equal(m.active_column_bursting[:, 0], o_tm.active_column_bursting[:, 0]), t
