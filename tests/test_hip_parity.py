"""GPU: the HIP path, driven through the Python drop-in classes and the C ABI, reproduces the
golden vectors of the unmodified reference and the oracle, bit for bit."""

import glob
import os

import numpy as np
import pytest

import golden_replay as gr

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRAJ = sorted(glob.glob(os.path.join(GOLDEN, "traj_*.npz")))


@pytest.mark.parametrize("path", TRAJ, ids=[os.path.basename(p)[5:-4] for p in TRAJ])
def test_hip_replays_reference_trajectory(path):
    from hip_impl import HipImpl
    g = gr.load(path)
    n = gr.replay(g, HipImpl(g))
    assert n == int(g["steps"])


def test_hip_vs_oracle_fresh_config_k32():
    from hip_impl import lockstep_vs_oracle
    lockstep_vs_oracle(seed=101, input_dim=1024, column_dim=4096, cell_dim=32, patterns=60, density=0.03,
                       noise=0.005, steps=330, store_every=40)


def test_hip_vs_oracle_learning_off_and_jumps():
    from hip_impl import lockstep_vs_oracle
    lockstep_vs_oracle(seed=102, input_dim=333, column_dim=2048, cell_dim=12, patterns=50, density=0.08,
                       noise=0.02, steps=300, store_every=50, jump=0.2, learning_schedule=lambda t: (t % 19) != 4)


def test_hip_small_slots_64():
    from hip_impl import lockstep_vs_oracle
    lockstep_vs_oracle(seed=103, input_dim=512, column_dim=4096, cell_dim=16, patterns=50, density=0.04,
                       noise=0.005, steps=260, store_every=50, segment_slots=64)


def test_harness_counters_match_reference():
    """example.py:50-57 counters (bursting / correct / incorrect columns) over 600 steps: equal to the
    reference's, and showing its learning curve."""
    from hip_impl import make_htm
    z = np.load(os.path.join(GOLDEN, "harness_example.npz"))
    seed, I, C, K, P = int(z["seed"]), int(z["input_dim"]), int(z["column_dim"]), int(z["cell_dim"]), int(z["patterns"])
    np.random.seed(seed)
    perm = np.random.randn(C, I) * 0.1 + 0.0
    htm = make_htm(I, C, K, round(C * 0.02), seed, perm)
    rng = np.random.RandomState(seed + 1)
    bank = rng.rand(P, I) < float(z["density"])
    got = []
    for t in range(int(z["steps"])):
        prev_col_pred = htm.temporal_memory.last_state.cell_prediction.max(axis=1)
        x = bank[t % P] ^ (rng.rand(I) < float(z["noise"]))
        sp_state, tm_state = htm.process(x)
        burst = int(tm_state.active_column_bursting.sum())
        correct = int(prev_col_pred[sp_state.active_column].sum())
        got.append((burst, correct, int(prev_col_pred.sum() - correct)))
    assert np.array_equal(np.array(got, dtype=np.int32), z["counters"])
    # the learning curve a user of example.py watches: everything bursts for four passes
    # (0.21 + 3 x 0.1 reaches the 0.5 threshold), then the sequence is predicted
    got = np.array(got).reshape(-1, P, 3).mean(axis=1)
    k = round(C * 0.02)
    assert (got[:4, 0] == k).all() and (got[:4, 1] == 0).all()
    assert got[4, 0] < 0.15 * k and got[5, 0] < 0.05 * k and got[5, 1] > 0.95 * k and got[:, 2].max() < 0.5


def test_full_size_65536x32_state_handoff_and_parity():
    """BASELINE.json configs[2] (65 536 columns x 32 cells, 1024 inputs, 2 % density): run the
    GPU through a learned phase with the batched entry point (hipGraph replay), hand the whole
    state to the oracle, then step both in lock-step and compare every output bit for bit.
    Also checks size-independent properties of the full-size state."""
    import bench
    from hip_impl import step_outputs
    from oracle import HTMOracle
    w = dict(bench.WORKLOAD)
    noisy, perm = bench.make_inputs(w)
    htm = bench.build_htm(w, perm, 0)
    eng = htm.engine
    C, I, K = w["column_dim"], w["input_dim"], w["cell_dim"]
    htm.run(noisy, 260, learning=True, use_graph=True)          # > 5 passes over the 50 patterns
    info = eng.check_capacity()
    assert info.step_index == 260 and info.segments > 50000
    # properties: k winners, sorted, unique; segcount is the histogram of seg_cell; rows packed
    sp = eng.read_sp_fields()
    assert len(sp["active_column"]) == htm.active_columns and np.all(np.diff(sp["active_column"]) > 0)
    thr = np.sort(sp["boosted_overlaps"])[-htm.active_columns]
    assert sp["boosted_overlaps"][sp["active_column"]].min() >= thr
    st = eng.read_store()
    assert np.array_equal(np.bincount(st["seg_cell"], minlength=C * K), st["segcount"])
    valid = st["presyn"] >= 0
    assert np.array_equal(valid.sum(axis=1), st["seg_nsyn"])
    assert np.all(valid[:, :-1] >= valid[:, 1:])                 # valid slots first
    assert (st["perm"][valid] >= 0).all() and (st["presyn"][valid] < C * K).all()
    # hand-off: GPU -> oracle, then 6 lock-step timesteps
    ora = HTMOracle(I, C, K, seed=0, permanence=eng.get_permanence())
    ora.spatial_pooler.duty_cycle = eng.read_duty_cycle().copy()
    ora.temporal_memory.import_state(eng.export_tm_state())
    for t in range(6):
        x = noisy[(260 + t) % len(noisy)]
        o_sp, o_tm = ora.step(x)
        h_sp, h_tm = htm.process(x)
        got = step_outputs(h_sp, h_tm, K)
        od = o_tm.distal_state
        want = dict(active_column=o_sp.active_column, overlaps=o_sp.overlaps, boosted=o_sp.boosted_overlaps,
                    bursting=o_tm.active_column_bursting[:, 0],
                    act_bits=np.packbits(o_tm.cell_activation.reshape(-1), bitorder="little"),
                    pred_bits=np.packbits(o_tm.cell_prediction.reshape(-1), bitorder="little"),
                    winner=o_tm.winner_cell[0] * K + o_tm.winner_cell[1], matching=od.matching_segment,
                    match_pot=od.segment_potential[od.matching_segment], match_act=od.matching_segment_activation,
                    match_active=od.matching_segment_active, S=len(od.segment_potential))
        for key in gr.FIELDS:
            a, b = np.asarray(got[key]), np.asarray(want[key])
            if key == "boosted":
                a, b = a.view(np.int64), b.view(np.int64)
            assert a.shape == b.shape and np.array_equal(a, b), f"full-size step {t}: {key}"
    # oracle -> GPU: import the oracle's state into a fresh engine and continue identically
    htm2 = bench.build_htm(w, ora.spatial_pooler.permanence, 0)
    htm2.engine.write(4, ora.spatial_pooler.duty_cycle, np.float32)          # HTM_F_DUTY_CYCLE
    htm2.engine.import_tm_state(ora.temporal_memory.export_state())
    x = noisy[(266) % len(noisy)]
    a_sp, a_tm = htm.process(x)
    b_sp, b_tm = htm2.process(x)
    for key in gr.FIELDS:
        u, v = np.asarray(step_outputs(a_sp, a_tm, K)[key]), np.asarray(step_outputs(b_sp, b_tm, K)[key])
        assert u.shape == v.shape and np.array_equal(u, v), f"import round trip: {key}"


@pytest.fixture(scope="module")
def full_size_oracle():
    """The NumPy oracle stepped 260 times FROM THE SEED at BASELINE.json configs[2] (65 536 x 32, bench.py's
    workload and input stream): an independent reference for the state the timed path must reach."""
    import bench
    from oracle import HTMOracle
    w = dict(bench.WORKLOAD)
    noisy, perm = bench.make_inputs(w)
    ora = HTMOracle(w["input_dim"], w["column_dim"], w["cell_dim"], seed=0, permanence=perm)
    n = 260                                                      # > 5 passes over the 50 patterns: bursting, then predicted
    for t in range(n):
        o_sp, o_tm = ora.step(noisy[t % len(noisy)])
    return dict(w=w, noisy=noisy, ora=ora, o_sp=o_sp, o_tm=o_tm, n=n)


@pytest.mark.parametrize("use_graph,pipeline", [(True, True), (False, True), (False, False)],
                         ids=["graph-pipelined", "eager-pipelined", "one-role-per-launch"])
def test_full_size_timed_path_equals_from_scratch_oracle(full_size_oracle, use_graph, pipeline):
    """The schedule bench.py times (pipelined launches, 16-step hipGraphs) and its two plainer forms, each run from
    scratch at the full size and compared with the from-scratch oracle: last step's outputs and the whole state."""
    import bench
    from hip_impl import compare_store_with_oracle, step_outputs
    f = full_size_oracle
    w, noisy, ora, n = f["w"], f["noisy"], f["ora"], f["n"]
    K = w["cell_dim"]
    _, perm = bench.make_inputs(w)
    htm = bench.build_htm(w, perm, 0)
    del perm
    htm.run(noisy, n, learning=True, use_graph=use_graph, pipeline=pipeline)
    eng = htm.engine
    info = eng.check_capacity()
    assert info.step_index == n and info.select_fallbacks == 0
    print(f"threshold bins cut to the k-th key's sub-bin: {info.select_zoom_steps} of {n} steps")
    sp_state = type(htm.spatial_pooler).State(eng, eng.steps)
    tm_state = htm.temporal_memory.last_state
    got = step_outputs(sp_state, tm_state, K)
    o_sp, o_tm = f["o_sp"], f["o_tm"]
    od = o_tm.distal_state
    want = dict(active_column=o_sp.active_column, overlaps=o_sp.overlaps, boosted=o_sp.boosted_overlaps,
                bursting=o_tm.active_column_bursting[:, 0],
                act_bits=np.packbits(o_tm.cell_activation.reshape(-1), bitorder="little"),
                pred_bits=np.packbits(o_tm.cell_prediction.reshape(-1), bitorder="little"),
                winner=o_tm.winner_cell[0] * K + o_tm.winner_cell[1], matching=od.matching_segment,
                match_pot=od.segment_potential[od.matching_segment], match_act=od.matching_segment_activation,
                match_active=od.matching_segment_active, S=len(od.segment_potential))
    for key in gr.FIELDS:
        a, b = np.asarray(got[key]), np.asarray(want[key])
        if key == "boosted":
            a, b = a.view(np.int64), b.view(np.int64)
        assert a.shape == b.shape and np.array_equal(a, b), f"{key} after {n} steps"
    d = tm_state.distal_state
    assert np.array_equal(d.segment_potential, od.segment_potential)
    assert np.array_equal(d.max_jittered_potential.view(np.int32), od.max_jittered_potential.view(np.int32))
    compare_store_with_oracle(n - 1, ora, htm)       # seg_cell, seg_nsyn, segcount, synapses + permanence bits, SP permanence, duty


def test_config1_16384_columns_sp_only():
    """BASELINE.json configs[1]: 16 384 columns, 2 % input sparsity, SpatialPooler only."""
    import bithtm_amd as B
    from oracle import SpatialPoolerOracle
    I, C = 1024, 16384
    k = round(C * 0.02)
    np.random.seed(0)
    sp = B.SpatialPooler(I, C, k)
    ora = SpatialPoolerOracle(I, C, k, permanence=sp.proximal_projection.permanence.copy())
    rng = np.random.RandomState(1)
    bank = rng.rand(50, I) < 0.02
    for t in range(150):
        x = bank[t % 50] ^ (rng.rand(I) < 0.005)
        got, want = sp.process(x), ora.step(x)
        assert np.array_equal(got.active_column, want.active_column), t
        assert np.array_equal(got.overlaps, want.overlaps), t
        assert np.array_equal(got.boosted_overlaps.view(np.int64), want.boosted_overlaps.view(np.int64)), t
    assert np.array_equal(sp.proximal_projection.permanence.view(np.int64), ora.permanence.view(np.int64))


def test_config4_shape_262144_columns_16_cells():
    """BASELINE.json configs[4] shape (262 144 columns x 16 cells; learned, not pre-populated): the
    largest grid of the count-in-emit path (1024 blocks) and a 32 KiB active-column bitmap in LDS."""
    from hip_impl import lockstep_vs_oracle
    lockstep_vs_oracle(seed=7, input_dim=1024, column_dim=262144, cell_dim=16, patterns=6, density=0.02,
                       noise=0.002, steps=26, store_every=25, segment_capacity=1 << 20)


def test_more_than_1024_column_blocks_uses_separate_count_kernel():
    """column_dim > 262 144: the per-block counts come from k_sp_count instead of the in-kernel exchange."""
    from hip_impl import lockstep_vs_oracle
    lockstep_vs_oracle(seed=8, input_dim=96, column_dim=327680, cell_dim=4, patterns=5, density=0.1,
                       noise=0.01, steps=22, store_every=21, segment_capacity=1 << 20)


def test_emit_grid_too_large_for_the_pipelined_launch_falls_back():
    """204 800 columns = 800 emit blocks: they fit k_sp_emit (select finished in-kernel) but not
    k_open_emit together with the activation blocks, so a batched run must not use the pipelined
    schedule -- the engine asks the runtime what is resident at once instead of assuming it.  The
    result has to be the oracle's either way."""
    from hip_impl import make_htm, compare_store_with_oracle
    from oracle import HTMOracle
    I, C, K, P, steps = 64, 204800, 4, 6, 20
    k = round(C * 0.02)
    np.random.seed(23)
    ora = HTMOracle(I, C, K, active_columns=k, seed=23)
    htm = make_htm(I, C, K, k, 23, ora.spatial_pooler.permanence.copy(), None, None, segment_capacity=1 << 19)
    bank = np.random.RandomState(24).rand(P, I) < 0.12
    for t in range(steps):
        o_sp, _ = ora.step(bank[t % P])
    htm.run(bank, steps - 4)
    htm.engine.profile(True)                        # the launches of a (requested) pipelined run, by name
    htm.run(bank, 4, use_graph=False, pipeline=True)
    names = set(htm.engine.profile_read())
    htm.engine.profile(False)
    assert "sp_emit" in names and "tm_activate+sp_emit" not in names, names
    assert np.array_equal(htm.engine.read_sp_fields()["active_column"], o_sp.active_column)
    compare_store_with_oracle(steps - 1, ora, htm)
    htm.engine.check_capacity()


@pytest.fixture(scope="module")
def large_pool_oracle():
    """The NumPy oracle stepped 260 times FROM THE SEED at 65 536 x 32 with bench.py's LARGE_POOL input stream (350
    patterns): everything bursts, 1 311 new segments per step -- 340 k segments, past the 294 912 above which the scan
    takes its streaming form (htm_engine.hip: scan_pool_is_large)."""
    import bench
    from oracle import HTMOracle
    w = dict(bench.LARGE_POOL)
    noisy, perm = bench.make_inputs(w)
    ora = HTMOracle(w["input_dim"], w["column_dim"], w["cell_dim"], seed=0, permanence=perm)
    n = 260
    for t in range(n):
        o_sp, o_tm = ora.step(noisy[t % len(noisy)])
    assert ora.temporal_memory.S > 320000
    return dict(w=w, noisy=noisy, ora=ora, o_sp=o_sp, o_tm=o_tm, n=n)


@pytest.mark.parametrize("launches", [2, 3, 4], ids=["two-launches", "three-launches", "four-launches"])
def test_full_size_pool_outgrows_the_resident_scan_in_mid_run(large_pool_oracle, launches, monkeypatch):
    """The headline shape fed 350 patterns instead of 50: the pool crosses the large-pool threshold in the middle of a
    stream of htm.run calls (graph keys change, the last launch becomes k_learn_scan_emit<E, 4, false>: the streaming scan
    beside the learning role and the select finish -- or k_scan_sel<*, 1> in the four-launch schedule).  From scratch
    against the from-scratch oracle: last step's outputs, every potential, the per-cell maxima, the whole state."""
    import bench
    from hip_impl import compare_store_with_oracle, step_outputs
    lean = launches < 4
    if launches != 2:                              # (two is the default)
        monkeypatch.setenv("BITHTM_LEAN", "1" if launches == 3 else "0")
    f = large_pool_oracle
    w, noisy, ora, n = f["w"], f["noisy"], f["ora"], f["n"]
    K = w["cell_dim"]
    _, perm = bench.make_inputs(w)
    htm = bench.build_htm(w, perm, 0)
    del perm
    eng = htm.engine
    bank = eng.upload_bank(noisy)
    plans = []
    for chunk, graph in ((120, True), (70, True), (40, True), (24, False)):       # S = 157 k, 249 k, 301 k (> 294 912), 333 k
        plans.append(eng.run_plan(chunk, use_graph=graph, continuing=True))
        eng.run(bank, len(noisy), chunk, use_graph=graph, continuing=True)
        eng.sync()                                   # (the call's last copy -- the segment count -- has landed: the next call sees it)
    assert [p["scan_large"] for p in plans] == [False, False, False, True] and all(p["lean"] == lean for p in plans), plans
    eng.profile(True)                                # the launches of the last steps, by name (eager under the profile)
    eng.run(bank, len(noisy), 6, continuing=False)
    names = set(eng.profile_read())
    eng.profile(False)
    assert ("tm_learn+tm_scan_large+sp_emit" if lean else "tm_scan_large+sp_select") in names, names
    assert ("tm_activate+tm_mid+sp_learn+sp_overlap" in names) == (launches == 2) and ("tm_mid+sp_overlap" in names) == (launches == 3), names
    assert not ({"tm_learn+tm_scan+sp_emit", "tm_scan+sp_select", "tm_scan"} & names), names
    info = eng.check_capacity()
    assert info.step_index == n and info.segments == ora.temporal_memory.S
    sp_state = type(htm.spatial_pooler).State(eng, eng.steps)
    tm_state = htm.temporal_memory.last_state
    got = step_outputs(sp_state, tm_state, K)
    o_sp, o_tm = f["o_sp"], f["o_tm"]
    od = o_tm.distal_state
    want = dict(active_column=o_sp.active_column, overlaps=o_sp.overlaps, boosted=o_sp.boosted_overlaps,
                bursting=o_tm.active_column_bursting[:, 0],
                act_bits=np.packbits(o_tm.cell_activation.reshape(-1), bitorder="little"),
                pred_bits=np.packbits(o_tm.cell_prediction.reshape(-1), bitorder="little"),
                winner=o_tm.winner_cell[0] * K + o_tm.winner_cell[1], matching=od.matching_segment,
                match_pot=od.segment_potential[od.matching_segment], match_act=od.matching_segment_activation,
                match_active=od.matching_segment_active, S=len(od.segment_potential))
    for key in gr.FIELDS:
        a, b = np.asarray(got[key]), np.asarray(want[key])
        if key == "boosted":
            a, b = a.view(np.int64), b.view(np.int64)
        assert a.shape == b.shape and np.array_equal(a, b), f"{key} after {n} steps"
    d = tm_state.distal_state
    assert np.array_equal(d.segment_potential, od.segment_potential)
    assert np.array_equal(d.max_jittered_potential.view(np.int32), od.max_jittered_potential.view(np.int32))
    compare_store_with_oracle(n - 1, ora, htm)


def test_full_size_large_learned_pool_against_the_oracle():
    """bench.py's `large_pool` leg as a test: 65 536 x 32 with 350 patterns brought to the learned state on the GPU (10 passes,
    S ~ 0.65 M segments: every step of the timed schedule runs the streaming scan beside the learning role and the select
    finish), the whole state handed to the oracle, and then the bench's own parity check -- the next timesteps called as the
    timed region calls them (streamed chunks, prepared graphs; one chunk long enough to replay hipGraphs), every Temporal
    Memory output at the chunk boundaries, the Spatial Pooler's outputs and the whole state at the end."""
    import bench
    from oracle import HTMOracle
    w = dict(bench.LARGE_POOL)
    noisy, perm = bench.make_inputs(w)
    htm = bench.build_htm(w, perm, 0)
    del perm
    eng = htm.engine
    bank = eng.upload_bank(noisy)
    n_bank = len(noisy)
    run = dict(learning=True, use_graph=True, pipeline=True)
    for _ in range(10):
        eng.run(bank, n_bank, w["patterns"], **run)
        eng.sync()
    info = eng.check_capacity()
    assert info.segments > 1000000 and info.select_fallbacks == 0, (info.segments, info.select_fallbacks)
    assert eng.run_plan(100, **run) == dict(hip_graph=True, pipelined=True, lean=True, scan_large=True)
    # size-independent properties of the learned state
    st = eng.read_store()
    C, K = w["column_dim"], w["cell_dim"]
    assert np.array_equal(np.bincount(st["seg_cell"], minlength=C * K), st["segcount"])
    valid = st["presyn"] >= 0
    assert np.array_equal(valid.sum(axis=1), st["seg_nsyn"]) and np.all(valid[:, :-1] >= valid[:, 1:])
    del st, valid
    ora = HTMOracle(w["input_dim"], C, K, seed=0, permanence=eng.get_permanence())
    ora.spatial_pooler.duty_cycle = eng.read_duty_cycle().copy()
    ora.temporal_memory.import_state(eng.export_tm_state())
    start = int(eng.info().step_index)
    states = [ora.step(noisy[(start + t) % n_bank]) for t in range(70)]
    out = bench.check_against_oracle(w, htm, noisy, bank, ora, states, run, chunk_plan=(5,))
    assert out["parity"] == "ok" and "65 (graph+large-pool scan)" in out["parity_checked"], out
    # the learning curve: after ten passes the sequence is predicted (few bursting columns)
    assert states[-1][1].active_column_bursting.mean() < 0.4
