"""GPU: the column-sharded HIP path.  R shards run inside one process on the one GPU of the test
box (the all-gather becomes device copies), and every rank's share of every result must equal the
unsharded oracle bit for bit -- the same check tests/test_sharded_gloo.py makes on the CPU protocol."""

import os

import numpy as np
import pytest

from oracle import HTMOracle, SPParams, TMParams, canonical_synapses

pytestmark = pytest.mark.gpu


def _run(world, I, C, K, P, density, noise, steps, jump, seed, spp=None, tmp=None, slots=128):
    import bithtm_amd as B
    from bithtm_amd.distributed import LocalGroup
    from bithtm_amd.engine import words_to_bool
    from bithtm_amd import _lib as L
    np.random.seed(seed)
    spp_, tmp_ = spp or SPParams(), tmp or TMParams()
    perm = np.random.randn(C, I) * spp_.permanence_std + spp_.permanence_mean
    k = round(C * 0.02)
    ora = HTMOracle(I, C, K, active_columns=k, seed=seed, sp_params=spp, tm_params=tmp, permanence=perm)

    def parts(r):
        prox = B.DenseProjection.__new__(B.DenseProjection)
        prox.input_dim, prox.output_dim = I, C
        prox.permanence_threshold, prox.permanence_increment, prox.permanence_decrement = \
            spp_.permanence_threshold, spp_.permanence_increment, spp_.permanence_decrement
        prox._engine, prox._permanence = None, perm
        return dict(proximal=prox,
                    boosting=B.ExponentialBoosting(C, k, intensity=spp_.boost_intensity, momentum=spp_.boost_momentum),
                    distal=B.PredictiveProjection(C * K, segment_slots=slots, **{f: getattr(tmp_, f) for f in tmp_.__dataclass_fields__}))
    group = LocalGroup(world, I, C, K, active_columns=k, make_parts=parts, seed=seed)
    rng = np.random.RandomState(seed + 1)
    bank = rng.rand(P, I) < density
    thr = tmp_.segment_matching_threshold
    dead_seen = 0
    for t in range(steps):
        idx = int(rng.randint(P)) if (jump > 0 and rng.rand() < jump) else t % P
        x = bank[idx] ^ (rng.rand(I) < noise)
        learning = (t % 23) != 5
        o_sp, o_tm = ora.step(x, learning=learning)
        group.process(x, learning=learning)
        check_store = (t % 25 == 0) or t == steps - 1
        otm = ora.temporal_memory
        for m in group.members:
            eng = m.engine
            c0, c1 = m.column_range
            info = eng.check_capacity()
            tag = f"rank {m.rank}/{world} step {t}"
            assert info.segments == otm.S, (f"{tag}: S {info.segments} vs {otm.S}; requests={info.new_segment_requests} "
                                            f"recycled={info.recycled_segments} appended={info.appended_segments} learning={learning}")
            act_cols = eng.read(L.F_ACTIVE_COLUMN, np.int32, k)
            assert np.array_equal(act_cols, o_sp.active_column), f"{tag}: active columns"
            act = words_to_bool(eng.read(L.F_CELL_ACTIVATION, np.uint32, C), K)
            assert np.array_equal(act, o_tm.cell_activation), f"{tag}: cell activation (replicated)"
            pred = words_to_bool(eng.read(L.F_CELL_PREDICTION, np.uint32, C), K)
            assert np.array_equal(pred[c0:c1], o_tm.cell_prediction[c0:c1]), f"{tag}: cell prediction (own)"
            win = eng.read(L.F_WINNER_CELL, np.int32, info.winner_cells)
            assert np.array_equal(win, o_tm.winner_cell[0] * K + o_tm.winner_cell[1]), f"{tag}: winner cells"
            burst = eng.read(L.F_BURSTING, np.uint8, k).astype(bool)
            assert np.array_equal(burst, o_tm.active_column_bursting[:, 0]), f"{tag}: bursting"
            boosted = eng.read(L.F_BOOSTED, np.float64, C)
            assert np.array_equal(boosted[c0:c1].view(np.int64), o_sp.boosted_overlaps[c0:c1].view(np.int64)), f"{tag}: boosted (own)"
            # the rank's rows hold exactly the segments its cells own, under their global ids
            rows = info.local_segments
            gid = eng.read(L.F_SEG_GID, np.int32, rows)
            live = np.flatnonzero(gid >= 0)
            owned = np.flatnonzero((otm.seg_cell[:otm.S] // K >= c0) & (otm.seg_cell[:otm.S] // K < c1))
            assert np.array_equal(np.sort(gid[live]), owned), f"{tag}: ids of the own segments"
            seg_cell = eng.read(L.F_SEG_CELL, np.int32, rows)
            assert np.array_equal(seg_cell[live], otm.seg_cell[gid[live]]), f"{tag}: seg_cell (own)"
            nsyn = eng.read(L.F_SEG_NSYN, np.int32, rows)
            assert np.array_equal(nsyn[live], otm.seg_nsyn[gid[live]]), f"{tag}: nsyn (own)"
            od = o_tm.distal_state
            mrow = eng.read(L.F_MATCH_SEGMENT, np.int32, rows)
            minfo = eng.read(L.F_MATCH_INFO, np.uint32, rows)
            order = np.argsort(gid[mrow], kind="stable")
            mine = np.isin(od.matching_segment, owned)
            assert np.array_equal(gid[mrow][order], od.matching_segment[mine]), f"{tag}: matching segments (own)"
            assert np.array_equal((minfo[order] >> 31).astype(bool), od.matching_segment_active[mine]), f"{tag}: active segments (own)"
            assert np.array_equal((minfo[order] & 0xFFF).astype(np.int64), od.segment_potential[od.matching_segment[mine]]), f"{tag}: potentials (own)"
            dead_seen += int((otm.seg_nsyn[:otm.S] < thr).sum() > 0)
            if check_store:
                E = eng.segment_slots
                presyn = eng.read(L.F_SEG_PRESYN, np.int32, rows * E).reshape(rows, E)
                pm = eng.read(L.F_SEG_PERM, np.float32, rows * E).reshape(rows, E)
                pot = eng.read(L.F_SEG_POTENTIAL, np.int32, rows)
                assert np.array_equal(pot[live].astype(np.int64), od.segment_potential[gid[live]]), f"{tag}: all potentials (own)"
                a = canonical_synapses(seg_cell[live], presyn[live], pm[live])
                b = canonical_synapses(otm.seg_cell[gid[live]], otm.presyn[gid[live]], otm.perm[gid[live]])
                for (ca, ia, pa), (cb, ib, pb) in zip(a, b):
                    assert ca == cb and np.array_equal(ia, ib) and np.array_equal(pa.view(np.int32), pb.view(np.int32)), \
                        f"{tag}: synapses of an owned segment"
                segcount = eng.read(L.F_SEGCOUNT, np.int32, C * K)
                assert np.array_equal(segcount[c0 * K:c1 * K], otm.segcount[c0 * K:c1 * K]), f"{tag}: segcount (own)"
                assert np.array_equal(eng.get_permanence(c0, c1 - c0).view(np.int64),
                                      ora.spatial_pooler.permanence[c0:c1].view(np.int64)), f"{tag}: SP permanence (own)"
                assert np.array_equal(eng.read_duty_cycle()[c0:c1].view(np.int32),
                                      ora.spatial_pooler.duty_cycle[c0:c1].view(np.int32)), f"{tag}: duty (own)"
    exact = max(m.engine.info().candidate_exact_steps for m in group.members)
    hot = min(m.engine.info().hot_select_steps for m in group.members)
    print(f"world {world}, {C} columns: local selects that cut their threshold bin exactly in {exact} of {steps} steps; "
          f"global selects settled among the hot lists in {hot}")
    _run.exact_steps, _run.hot_steps = exact, hot
    return dead_seen


def test_rccl_all_gather_through_the_library_at_world_size_one():
    """htm_shard_step's exchange is ncclAllGather, called from inside the library on the handle's stream (librccl is
    loaded at run time).  One GPU can hold one rank: the self-test creates a communicator of size 1 and moves one
    record through it."""
    from bithtm_amd import _lib
    lib = _lib.load()
    rc = lib.htm_rccl_selftest(0)
    assert rc in (0, 1), lib.htm_last_error(None)      # 0: the collective also replays from a captured hipGraph (htm_shard_run's form)
    print("RCCL all-gather captured into a hipGraph and replayed" if rc == 0 else "RCCL all-gather works eagerly; NOT capturable on this runtime")


def test_two_shards_default_parameters():
    _run(2, I=256, C=2048, K=32, P=50, density=0.06, noise=0.01, steps=300, jump=0.0, seed=51)


@pytest.mark.parametrize("env", [{"BITHTM_SHARD_WINDOW": "0"}, {"BITHTM_FUSE_TM": "0"}, {"BITHTM_SEL_WINDOW_OFFSET": "4000"},
                                 {"BITHTM_CAND_TAKE_ALL": "0"}],
                         ids=lambda e: ",".join(f"{k[7:]}={v}" for k, v in e.items()))
def test_four_shards_with_the_other_select_and_launch_forms(env, monkeypatch):
    """The sharded step with two launched digits instead of the windowed histogram in the local select, with the
    learning role and the scan as two launches, with both windows (local and global) forced to miss every step, and with a
    local select that always cuts its threshold bin exactly (the record exchange) instead of handing the bin over."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    _run(4, I=200, C=1024, K=16, P=20, density=0.08, noise=0.01, steps=120, jump=0.02, seed=57)
    if env.get("BITHTM_CAND_TAKE_ALL") == "0":
        assert _run.exact_steps >= 118              # (every step but the first: that one has no window yet)


def test_four_shards_stress_parameters_with_recycling():
    spp = SPParams(permanence_mean=0.01, permanence_std=0.08, permanence_threshold=0.02, permanence_increment=0.05,
                   permanence_decrement=0.02, boost_intensity=0.5, boost_momentum=0.95)
    tmp = TMParams(permanence_initial=0.3, permanence_threshold=0.45, permanence_increment=0.12, permanence_decrement=0.14,
                   permanence_punishment=0.2, segment_activation_threshold=12, segment_matching_threshold=9,
                   segment_sampling_synapses=20)
    dead = _run(4, I=256, C=1024, K=4, P=40, density=0.12, noise=0.02, steps=320, jump=0.15, seed=13, spp=spp, tmp=tmp)
    assert dead > 0        # dead segments existed, so cross-rank recycling was exercised


def test_eight_shards_k16():
    _run(8, I=512, C=4096, K=16, P=50, density=0.04, noise=0.005, steps=200, jump=0.0, seed=9)


def test_config3_65536_columns_eight_shards():
    """BASELINE.json configs[3]: 65 536 columns x 32 cells sharded 8-way (8 192 columns per shard),
    here with the 8 shards emulated on one GPU; every shard must agree with the unsharded oracle."""
    _run(8, I=1024, C=65536, K=32, P=12, density=0.02, noise=0.005, steps=60, jump=0.0, seed=0)
    assert _run.exact_steps <= 30               # the whole-bin hand-over is the usual path at this size


def test_random_sharded_configurations():
    """A seeded sample of shard counts, cell counts and thresholds (low thresholds make segments die and
    get recycled across ranks early)."""
    rng = np.random.RandomState(int(os.environ.get("BITHTM_SHARD_FUZZ_SEED", "77")))      # (a longer manual sweep: set seed and count)
    for _ in range(int(os.environ.get("BITHTM_SHARD_FUZZ_CONFIGS", "6"))):
        world = int(rng.choice([2, 4, 8]))
        C = int(rng.choice([1024, 2048, 4096]))
        K = int(rng.choice([4, 8, 16, 32]))
        tmp = None
        if rng.rand() < 0.6:
            thr = int(rng.choice([3, 5]))
            tmp = TMParams(segment_activation_threshold=thr, segment_matching_threshold=thr - int(rng.randint(0, 2)),
                           segment_sampling_synapses=int(rng.choice([8, 16])), permanence_punishment=float(rng.choice([0.01, 0.3])))
        _run(world, I=int(rng.choice([64, 200])), C=C, K=K, P=int(rng.choice([5, 12])), density=float(rng.choice([0.1, 0.3])),
             noise=float(rng.choice([0.0, 0.02])), steps=60, jump=float(rng.choice([0.0, 0.2])), seed=int(rng.randint(1 << 16)), tmp=tmp)


def test_one_candidate_per_rank():
    """active_columns = 1: every rank's record holds ONE candidate (the candidate -> (rank, entry) division of
    k_shard_select with KL = 1)."""
    import bithtm_amd as B
    from bithtm_amd.distributed import LocalGroup
    from bithtm_amd import _lib as L
    I, C, K, world, seed = 64, 256, 4, 2, 5
    np.random.seed(seed)
    perm = np.random.randn(C, I) * 0.1
    ora = HTMOracle(I, C, K, active_columns=1, seed=seed, permanence=perm)
    group = LocalGroup(world, I, C, K, active_columns=1, permanence=perm, seed=seed)
    rng = np.random.RandomState(seed + 1)
    bank = rng.rand(7, I) < 0.2
    for t in range(60):
        o_sp, o_tm = ora.step(bank[t % 7])
        group.process(bank[t % 7])
        for m in group.members:
            assert np.array_equal(m.engine.read(L.F_ACTIVE_COLUMN, np.int32, 1), o_sp.active_column), (t, m.rank)
            info = m.engine.check_capacity()
            assert info.segments == ora.temporal_memory.S, (t, m.rank)
            assert np.array_equal(m.engine.read(L.F_WINNER_CELL, np.int32, info.winner_cells), o_tm.winner_cell[0] * K + o_tm.winner_cell[1]), (t, m.rank)


@pytest.mark.parametrize("world", [2, 8])
def test_state_import_into_a_sharded_group_and_merged_export(world):
    """A learned state of an unsharded run handed to a column-sharded group (every rank is given the whole state and keeps
    the rows of its own cells' segments, under their global ids, with the replicated dead bits and recyclable counts), the
    group stepped on, and its merged export against the unsharded run stepped the same way -- recycling included."""
    from hip_impl import make_htm
    from bithtm_amd.distributed import LocalGroup
    import bithtm_amd as B
    I, C, K, P, seed = 160, 2048, 8, 14, 33
    k = round(C * 0.02)
    tmp = TMParams(permanence_punishment=0.2, permanence_decrement=0.14, segment_activation_threshold=9, segment_matching_threshold=7,
                   segment_sampling_synapses=18)
    np.random.seed(seed)
    perm = np.random.randn(C, I) * 0.1
    solo = make_htm(I, C, K, k, seed, perm, None, tmp)
    rng = np.random.RandomState(seed + 1)
    bank = rng.rand(P, I) < 0.1
    steps = [bank[int(rng.randint(P))] if rng.rand() < 0.15 else bank[t % P] for t in range(260)]
    for x in steps[:160]:
        solo.process(x)
    state = solo.state_dict()

    def parts(r):
        prox = B.DenseProjection.__new__(B.DenseProjection)
        prox.input_dim, prox.output_dim = I, C
        prox.permanence_threshold, prox.permanence_increment, prox.permanence_decrement = 0.0, 0.03, 0.015
        prox._engine, prox._permanence = None, np.zeros((C, I))
        return dict(proximal=prox, distal=B.PredictiveProjection(C * K, **{f: getattr(tmp, f) for f in tmp.__dataclass_fields__}))
    group = LocalGroup(world, I, C, K, active_columns=k, make_parts=parts, seed=seed)
    group.import_state({key[3:]: v for key, v in state.items() if key.startswith("tm_")}, state["sp_permanence"], state["sp_duty_cycle"])
    for x in steps[160:]:
        solo.process(x)
        group.process(x)
    for e in group.engines:
        e.check_capacity()
    want, got = solo.engine.export_tm_state(), group.export_tm_state()
    assert set(want) == set(got)
    assert (want["seg_nsyn"] < 7).any() and int(want["S"]) > int(state["tm_S"])       # deaths and growth after the import
    for key in want:
        a, b = np.asarray(got[key]), np.asarray(want[key])
        if key in ("perm", "max_jittered_potential", "matching_segment_jittered_potential"):
            a, b = a.view(np.int32), b.view(np.int32)
        if key in ("presyn", "perm"):                  # (slot order within a row is free)
            order_a, order_b = np.argsort(np.asarray(got["presyn"]), axis=1, kind="stable"), np.argsort(np.asarray(want["presyn"]), axis=1, kind="stable")
            a, b = np.take_along_axis(a, order_a, axis=1), np.take_along_axis(b, order_b, axis=1)
        assert a.shape == b.shape and np.array_equal(a, b), key
    for e in group.engines:
        c0, c1 = e.column_range
        assert np.array_equal(e.get_permanence(c0, c1 - c0).view(np.int64), solo.engine.get_permanence(c0, c1 - c0).view(np.int64))
        assert np.array_equal(e.read_duty_cycle()[c0:c1].view(np.int32), solo.engine.read_duty_cycle()[c0:c1].view(np.int32))


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_runs_inside_the_library_equal_step_by_step(world):
    """htm_shard_group_run (the loop inside the library: whole timesteps replayed as hipGraphs, the next step's overlap riding
    in the last launch of the current one) == one htm_shard_group_step call per timestep == the oracle, for run lengths
    that exercise every plan (first / steady / last steps, odd and even graph spans, eager)."""
    import bithtm_amd as B
    from bithtm_amd.distributed import LocalGroup
    from bithtm_amd import _lib as L
    I, C, K, P, seed = 200, 4096, 8, 13, 44
    k = round(C * 0.02)
    np.random.seed(seed)
    perm = np.random.randn(C, I) * 0.1
    ora = HTMOracle(I, C, K, active_columns=k, seed=seed, permanence=perm)
    bank = np.random.RandomState(seed + 1).rand(P, I) < 0.1
    lengths = [1, 2, 3, 40, 5, 17, 18, 1, 34]
    groups = {}
    for mode in ("graph", "eager", "nopipe", "stepwise"):
        g = LocalGroup(world, I, C, K, active_columns=k, permanence=perm, seed=seed)
        g.upload_bank(bank)
        for n in lengths:
            if mode == "stepwise":
                g.run(n, stepwise=True)
            else:
                g.run(n, use_graph=mode == "graph", pipeline=mode != "nopipe")
        groups[mode] = g
    for t in range(sum(lengths)):
        o_sp, o_tm = ora.step(bank[t % P])
    want = None
    for mode, g in groups.items():
        for e in g.engines:
            info = e.check_capacity()
            assert info.step_index == sum(lengths) and info.segments == ora.temporal_memory.S, mode
            assert np.array_equal(e.read(L.F_ACTIVE_COLUMN, np.int32, k), o_sp.active_column), mode
        got = g.export_tm_state()
        if want is None:
            want = got
            assert np.array_equal(got["prev_prediction"], o_tm.cell_prediction) and np.array_equal(got["seg_nsyn"], ora.temporal_memory.seg_nsyn[:ora.temporal_memory.S])
            continue
        for key in want:
            a, b = np.asarray(got[key]), np.asarray(want[key])
            if a.dtype.kind == "f":
                a, b = a.view(np.int32 if a.itemsize == 4 else np.int64), b.view(np.int32 if b.itemsize == 4 else np.int64)
            assert a.shape == b.shape and np.array_equal(a, b), (mode, key)
