"""CPU: host-side logic and the C-ABI surface (no compute calls, no GPU needed)."""

import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from bithtm_amd import _lib
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "bithtm_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:const\s+)?[A-Za-z_][A-Za-z0-9_\s\*]*?\b(htm_[a-z_]+)\s*\(", header, flags=re.M))
    declared -= {"htm_handle", "htm_status", "htm_config", "htm_info", "htm_field"}
    assert len(declared) >= 18
    from bithtm_amd import _lib
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.htm_abi_version() == _lib.ABI_VERSION == 4


def test_the_three_launches_keep_their_register_budget():
    """k_learn_scan_emit is three roles in one launch, all resident at once at 6 waves per SIMD: 80 VGPRs and NO scratch.
    One spilled register costs every role of the launch microseconds (measured: 32 bytes of scratch per lane, the scan's
    last block 7.8 -> 10.5 us, 37 -> 32.5 k timesteps/s; LABNOTES.md).  The compiler's own report of the build."""
    from bithtm_amd.build import kernel_resources
    res = kernel_resources()
    if res is None:
        pytest.skip("the library in the tree was not built here")
    timed = {k: v for k, v in res.items() if re.search(r"k_learn_scan_emitILi\dELi6E", k)}
    assert len(timed) == 8, sorted(res)
    for name, r in {**timed, **{k: v for k, v in res.items() if re.search(r"k_act_rows|k_mid_overlap|k_learn_scan_(tail|front)", k)}}.items():
        assert r["scratch_bytes_per_lane"] == 0, (name, r)
    for name, r in timed.items():
        assert r["vgprs"] <= 80 and r["occupancy"] >= 6, (name, r)
    # ... and the first launch of the default, two-launch schedule to 8 blocks per CU (every block of the bench shape resident at once)
    # with no more scratch than it was measured with (24 bytes per lane, all of them the middle role's: the same role costs k_mid_rows 20)
    first = [v for k, v in res.items() if "k_act_mid_rows" in k]
    assert len(first) == 1 and first[0]["occupancy"] >= 8 and first[0]["scratch_bytes_per_lane"] <= 24, first


def test_create_rejects_bad_config_without_touching_a_gpu(lib):
    import ctypes as C
    from bithtm_amd import _lib
    cfg = _lib.HtmConfig()
    h = C.c_void_p()
    assert lib.htm_create(C.byref(cfg), C.byref(h)) == -1          # struct_bytes mismatch
    assert b"size mismatch" in lib.htm_last_error(None)
    cfg.struct_bytes = C.sizeof(_lib.HtmConfig)
    cfg.column_dim, cfg.active_columns, cfg.enable_tm, cfg.cell_dim = 64, 4, 1, 65
    cfg.segment_slots, cfg.segment_capacity, cfg.segment_sampling_synapses = 128, 16, 32
    assert lib.htm_create(C.byref(cfg), C.byref(h)) == -1
    assert b"cell_dim" in lib.htm_last_error(None)


def test_pack_and_unpack_bits():
    from bithtm_amd.engine import pack_bits, words_to_bool, bool_to_words
    rng = np.random.RandomState(0)
    x = rng.rand(1000) < 0.3
    w = pack_bits(x, 32)
    assert w.dtype == np.uint32 and w.shape == (32,)
    for i in (0, 1, 31, 32, 33, 999):
        assert bool((w[i >> 5] >> (i & 31)) & 1) == bool(x[i])
    assert w[31] >> 8 == 0                      # bits beyond input_dim are zero
    m = rng.rand(50, 13) < 0.5
    assert np.array_equal(words_to_bool(bool_to_words(m), 13), m)


def test_config_scalars_equal_the_oracles_derivation():
    """The binding forms the implicit scalars with the reference's own Python expressions."""
    import inspect
    from bithtm_amd import engine
    from oracle.htm_oracle import SPParams, TMParams, sp_derived, tm_derived
    src = inspect.getsource(engine.Engine.__init__)
    for needle in ("1.0 * (inc + dec) - dec", "0.0 * (inc + dec) - dec", "np.float32(-(boosting.intensity / density))",
                   "np.float32(1.0 - boosting.momentum)", "1.0 * (a - b) + b", "0.0 * (a - b) + b"):
        assert needle in src
    d = sp_derived(SPParams(), 2048, 41)
    assert d.delta_on == 1.0 * (0.03 + 0.015) - 0.015 and d.delta_off == -0.015
    t = tm_derived(TMParams())
    assert t.learn_active == 0.1 and t.learn_inactive == -0.1 and t.punish_active == -0.01 and t.punish_inactive == 0.0


def test_plugin_arguments_are_type_checked():
    import bithtm_amd as B

    class Foreign:
        pass
    with pytest.raises(TypeError):
        B.SpatialPooler(10, 64, 2, boosting=Foreign())
    with pytest.raises(TypeError):
        B.TemporalMemory(64, 4, distal_projection=Foreign())
    sp = B.SpatialPooler(10, 64, 2, proximal_projection=B.DenseProjection(10, 64, permanence_threshold=0.1))
    assert sp.proximal_projection.permanence.shape == (64, 10)
    assert B.SpatialPooler.compute is B.SpatialPooler.process
    assert B.HierarchicalTemporalMemory.compute is B.HierarchicalTemporalMemory.process


def test_dense_projection_draws_like_the_reference():
    """projections.py:16: same expression => same matrix under the same seed."""
    import bithtm_amd as B
    np.random.seed(5)
    want = np.random.randn(32, 20) * 0.1 + 0.0
    np.random.seed(5)
    got = B.DenseProjection(20, 32).permanence
    assert np.array_equal(want, got)


def test_device_twins_match_the_oracle_sources():
    """The constants of the device exp / RNG headers are the oracle's."""
    from oracle import fexp
    h = open(os.path.join(ROOT, "bithtm_amd", "csrc", "htm_fexp.h")).read()
    for c in (fexp.LOG2E, fexp.LN2_HI, fexp.LN2_LO) + fexp.TAYLOR:
        assert c.hex().replace("0x1.", "0x1.") in h, c.hex()
    r = open(os.path.join(ROOT, "bithtm_amd", "csrc", "htm_rng.h")).read()
    for c in ("0x7FEB352Du", "0x846CA68Bu", "0x9E3779B9u"):
        assert c in r


def test_host_side_keyed_draws_equal_the_oracles():
    """bithtm_amd/_keyed.py (the draws a host-orchestrated TemporalMemory uses) is the oracle's / the device's generator."""
    from bithtm_amd._keyed import draw_unit, STREAM_LEAST_USED
    from oracle import keyed_rng
    rng = np.random.RandomState(5)
    a = rng.randint(0, 1 << 22, size=(7, 32))
    for seed, step in ((0, 0), (7, 123), (0xFFFFFFFF, 4_000_000_000), (12345, 999)):
        assert STREAM_LEAST_USED == keyed_rng.STREAM_LEAST_USED
        assert np.array_equal(draw_unit(seed, STREAM_LEAST_USED, step, a), keyed_rng.draw_unit(seed, keyed_rng.STREAM_LEAST_USED, step, a))
        assert np.array_equal(draw_unit(seed, 3, step, a, 17), keyed_rng.draw_unit(seed, 3, step, a, 17))


def test_temporal_memory_around_a_host_side_projection_needs_no_device():
    """TemporalMemory(distal_projection=<an object with the reference's interface>) orchestrates networks.py:91-128 on the
    host: with a projection that lives there too nothing touches the GPU -- the whole path runs (and is compared with the
    oracle's own step) in the CPU suite.  Likewise HierarchicalTemporalMemory with two foreign layers."""
    import bithtm_amd as B
    from types import SimpleNamespace
    from oracle import HTMOracle, SpatialPoolerOracle, TemporalMemoryOracle, TMParams
    C, K, k, seed = 256, 8, 12, 5

    class Projection:                               # the oracle's learning and scan behind the reference's PredictiveProjection interface
        def __init__(self):
            self.o = TemporalMemoryOracle(C, K, TMParams(segment_activation_threshold=5, segment_matching_threshold=4), seed)
            self.segment_matching_threshold = 4

        bundle_segments = property(lambda self: self.o.segcount)

        def get_jittered_potential_info(self, state, matching_segment_bundle=None):
            return state.max_jittered_potential, state.matching_segment_jittered_potential

        def process(self, active_input, return_jittered_potential_info=True):
            act = np.zeros(C * K, dtype=np.bool_)
            act[active_input] = True
            d = self.o._scan(act.reshape(C, K), self.o.step_index)
            self.o.step_index += 1
            return d

        def update(self, prev_state, input_activation, learning_output, output_punishment, winner_input=None, output_learning=None, epsilon=1e-8):
            if prev_state is None:
                return
            o = self.o
            o.prev_distal, o.prev_activation, o.prev_winner = prev_state, np.asarray(input_activation).reshape(C, K), winner_input
            o._learn(np.asarray(learning_output, dtype=np.int64), np.flatnonzero(~np.asarray(output_punishment).reshape(C, K).any(axis=1)), o.step_index)

    tm = B.TemporalMemory(C, K, distal_projection=Projection(), seed=seed)
    ora = TemporalMemoryOracle(C, K, TMParams(segment_activation_threshold=5, segment_matching_threshold=4), seed)
    rng = np.random.RandomState(6)
    seqs = [np.sort(rng.choice(C, k, replace=False)) for _ in range(6)]
    for t in range(60):
        cols = seqs[t % 6]
        want = ora.step(cols, learning=t % 7 != 3)
        got = tm.process(SimpleNamespace(active_column=cols[rng.permutation(k)]), learning=t % 7 != 3)
        assert np.array_equal(got.cell_prediction, want.cell_prediction) and np.array_equal(got.cell_activation, want.cell_activation), t
        assert np.array_equal(got.winner_cell[0] * K + got.winner_cell[1], want.winner_cell[0] * K + want.winner_cell[1]), t
        assert np.array_equal(got.distal_state.matching_segment, want.distal_state.matching_segment), t
    assert tm._engine is None and ora.S > 20 and np.array_equal(tm.distal_projection.o.seg_nsyn[:ora.S], ora.seg_nsyn[:ora.S])
    with pytest.raises(TypeError):
        B.TemporalMemory(C, K, distal_projection=object())

    class Layer:
        def __init__(self, step):
            self.step = step

        def process(self, x, learning=True):
            return self.step(x, learning)
    I = 64
    np.random.seed(seed)
    full = HTMOracle(I, C, K, active_columns=k, seed=seed)
    sp_o = SpatialPoolerOracle(I, C, k, permanence=full.spatial_pooler.permanence.copy())
    tm_o = TemporalMemoryOracle(C, K, seed=seed)
    htm = B.HierarchicalTemporalMemory(I, C, K, active_columns=k, spatial_pooler=Layer(lambda x, learning: sp_o.step(x, learning=learning)),
                                       temporal_memory=Layer(lambda s, learning: tm_o.step(s.active_column, learning=learning)))
    bank = rng.rand(5, I) < 0.2
    for t in range(30):
        o_sp, o_tm = full.step(bank[t % 5])
        s, m = htm.process(bank[t % 5])
        assert np.array_equal(s.active_column, o_sp.active_column) and np.array_equal(m.cell_prediction, o_tm.cell_prediction), t
    assert htm.engine is None
    with pytest.raises(RuntimeError):
        htm.state_dict()


def test_merging_the_ranks_exports():
    """distributed.merge_shard_states: local rows under global ids, own cells and own columns from every rank, into the
    state an unsharded handle exports."""
    from bithtm_amd.distributed import merge_shard_states
    C, K, E, S = 8, 2, 4, 5
    owner = np.array([0, 9, 3, 14, 8], dtype=np.int32)            # flat owner cell of segment id 0..4 (column = cell // K)
    rng = np.random.RandomState(1)
    presyn = rng.randint(0, C * K, size=(S, E)).astype(np.int32)
    perm = rng.rand(S, E).astype(np.float32)
    nsyn = np.array([4, 4, 2, 4, 1], dtype=np.int32)
    pot = np.array([3, 0, 2, 4, 1], dtype=np.int64)
    pred = rng.rand(C, K) < 0.5
    cmax = rng.rand(C * K).astype(np.float32)
    match = np.array([0, 3], dtype=np.int64)
    parts = []
    for r, (c0, c1) in enumerate(((0, 4), (4, 8))):
        mine = np.flatnonzero((owner // K >= c0) & (owner // K < c1))[::-1]      # local rows in some other order ...
        gid = np.concatenate([mine[:1], [-1], mine[1:]]).astype(np.int32)        # ... with a free row in between
        live = gid >= 0
        rows = lambda a, fill: np.where(live.reshape((-1,) + (1,) * (a.ndim - 1)), a[np.maximum(gid, 0)], fill)
        local_match = np.array([int(np.flatnonzero(gid == g)[0]) for g in match if g in gid], dtype=np.int64)
        parts.append(dict(S=np.int64(S), slots=np.int64(E), step_index=np.int64(7), seg_gid=gid, column_range=np.array([c0, c1]),
                          seg_cell=rows(owner, 0), seg_nsyn=rows(nsyn, 0), presyn=rows(presyn, -1), perm=rows(perm, -1.0),
                          segcount=np.where((np.arange(C * K) // K >= c0) & (np.arange(C * K) // K < c1), np.bincount(owner, minlength=C * K), 99).astype(np.int32),
                          prev_prediction=np.where(((np.arange(C) >= c0) & (np.arange(C) < c1))[:, None], pred, ~pred),
                          prev_activation=pred.copy(), prev_winner=np.array([1, 9], dtype=np.int64), has_prev_winner=np.bool_(True), has_distal=np.bool_(True),
                          segment_potential=rows(pot, 0), matching_segment=local_match,
                          matching_segment_activation=pot[gid[local_match]], matching_segment_active=pot[gid[local_match]] >= 4,
                          matching_segment_jittered_potential=pot[gid[local_match]].astype(np.float32) + 0.5,
                          max_jittered_potential=np.where((np.arange(C * K) // K >= c0) & (np.arange(C * K) // K < c1), cmax, -1.0).astype(np.float32),
                          prediction=np.where((np.arange(C * K) // K >= c0) & (np.arange(C * K) // K < c1), 1.0, 7.0)))
    out = merge_shard_states(parts[::-1], C, K)                                  # (any order of the parts)
    assert int(out["S"]) == S and np.array_equal(out["seg_cell"], owner) and np.array_equal(out["seg_nsyn"], nsyn)
    assert np.array_equal(out["presyn"], presyn) and np.array_equal(out["perm"], perm)
    assert np.array_equal(out["segcount"], np.bincount(owner, minlength=C * K)) and np.array_equal(out["prev_prediction"], pred)
    assert np.array_equal(out["segment_potential"], pot) and np.array_equal(out["max_jittered_potential"], cmax)
    assert np.array_equal(out["matching_segment"], match) and np.array_equal(out["matching_segment_activation"], pot[match])
    assert np.array_equal(out["matching_segment_active"], pot[match] >= 4) and (out["prediction"] == 1.0).all()
    parts[0]["seg_gid"] = parts[0]["seg_gid"].copy()
    parts[0]["seg_gid"][parts[0]["seg_gid"] >= 0] = parts[1]["seg_gid"][parts[1]["seg_gid"] >= 0][0]      # an id owned twice
    with pytest.raises(AssertionError):
        merge_shard_states(parts, C, K)
