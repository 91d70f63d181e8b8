"""Driver of the sanitizer build (tests/test_host_sanitizers.py runs it in a child process with the AddressSanitizer runtime
preloaded and BITHTM_LIBRARY pointing at the build): the engine's HOST code -- the 2 400 lines of C++ behind the C ABI --
run over a HIP runtime made of host memory (tests/host_stub/hip_stub_runtime.cpp: kernels do nothing, every copy is a
bounds-checked memcpy).  What must hold without a device: argument checks, launch plans and graph caches, state import /
export round trips (unsharded and column-sharded, 32 and 64 cell slots per column), read-back conversions."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bithtm_amd as B  # noqa: E402
from bithtm_amd import _lib as L  # noqa: E402
from bithtm_amd.distributed import merge_shard_states  # noqa: E402
from bithtm_amd.engine import Engine, HtmError  # noqa: E402
from oracle import TMParams, TemporalMemoryOracle  # noqa: E402


def oracle_state(C, K, k, steps, seed):
    tmp = TMParams(segment_activation_threshold=6, segment_matching_threshold=5, segment_sampling_synapses=12, permanence_punishment=0.1)
    ora = TemporalMemoryOracle(C, K, tmp, seed=seed)
    rng = np.random.RandomState(seed)
    seqs = [np.sort(rng.choice(C, k, replace=False)) for _ in range(5)]
    for t in range(steps):
        ora.step(seqs[t % 5])
    return tmp, ora.export_state()


def same_state(a, b, what):
    for key in ("S", "seg_cell", "seg_nsyn", "segcount", "prev_prediction", "prev_activation", "prev_winner", "has_prev_winner", "has_distal",
                "matching_segment", "matching_segment_activation", "matching_segment_active"):
        assert np.array_equal(np.asarray(a[key]), np.asarray(b[key])), (what, key)
    for key in ("matching_segment_jittered_potential", "max_jittered_potential"):
        assert np.array_equal(np.asarray(a[key], np.float32).view(np.int32), np.asarray(b[key], np.float32).view(np.int32)), (what, key)
    S, w = int(a["S"]), min(np.asarray(a["presyn"]).shape[1], np.asarray(b["presyn"]).shape[1])

    def canon(st):                                   # valid synapses first, in slot order
        presyn, perm = np.asarray(st["presyn"]).reshape(S, -1), np.asarray(st["perm"], np.float32).reshape(S, -1)
        order = np.argsort(presyn < 0, axis=1, kind="stable")
        return np.take_along_axis(presyn, order, axis=1)[:, :w], np.take_along_axis(perm, order, axis=1)[:, :w].view(np.int32)
    (pa, ma), (pb, mb) = canon(a), canon(b)
    assert np.array_equal(pa, pb) and np.array_equal(np.where(pa >= 0, ma, 0), np.where(pb >= 0, mb, 0)), (what, "synapses")


def distal(C, K, tmp, **kw):
    return B.PredictiveProjection(C * K, segment_slots=64, **{f: getattr(tmp, f) for f in tmp.__dataclass_fields__}, **kw)


def main():
    lib = L.load()
    rng = np.random.RandomState(1)
    # ---- whole models: every entry point of a fused handle, both cell layouts
    for K in (8, 40):
        I, C = 100, 512
        np.random.seed(K)
        htm = B.HierarchicalTemporalMemory(I, C, K)
        eng = htm.engine
        bank = rng.rand(7, I) < 0.1
        for x in bank:
            sp, tm = htm.process(x)
            assert tm.cell_prediction.shape == (C, K) and sp.overlaps.shape == (C,)
        dev = eng.upload_bank(bank)
        for n, g, pipe, cont in ((1, True, True, False), (2, True, True, True), (40, True, True, True), (17, False, True, True), (70, True, True, False),
                                 (5, True, False, False), (33, False, False, False)):
            eng.prepare(dev, len(bank), n, use_graph=g, pipeline=pipe, continuing=cont)
            eng.run(dev, len(bank), n, use_graph=g, pipeline=pipe, continuing=cont)
            eng.run_plan(n, use_graph=g, pipeline=pipe, continuing=cont)
        eng.profile(True)
        eng.run(dev, len(bank), 6)
        assert eng.profile_read()
        eng.profile(False)
        eng.set_epsilon(0.25)
        htm.process(bank[0], learning=False)
        info = eng.check_capacity()
        assert info.step_index == eng.steps
        eng.read_sp_fields(), eng.read_store(), eng.read_duty_cycle(), eng.get_permanence()
        st = htm.state_dict()
        htm.load_state_dict(st)
        htm.grow_pool()
        htm.process(bank[1])
    # ---- state hand-over through the C++ import / export, unsharded and column-sharded
    for K in (8, 40):
        C, k = 512, 10
        tmp, st = oracle_state(C, K, k, 45, seed=K)
        assert int(st["S"]) >= 50 and len(st["matching_segment"]) > 0
        eng = Engine(0, C, K, k, distal=distal(C, K, tmp), seed=3)
        eng.import_tm_state(st)
        out = eng.export_tm_state()
        same_state(st, out, f"unsharded K={K}")
        assert np.array_equal(eng.read_rows(L.F_SEG_NSYN, np.int32, 5, 20), st["seg_nsyn"][5:25])
        # PredictiveProjection on its own: update with and without a punishment mask, scan
        wpc = eng.cell_words // C
        cols = np.arange(0, 40, 4)
        ww = rng.randint(1, 200, size=(len(cols), wpc)).astype(np.uint32)
        eng.tm_update(cols, ww, ww & 5, None)
        eng.tm_update(cols, ww, ww & 5, np.zeros(C * wpc, np.uint32))
        eng.tm_scan(np.zeros(C * wpc, np.uint32))
        eng.tm_step(cols, learning=True)
        if K <= 32:
            for world in (2, 4):
                perm = np.zeros((C, 64))
                parts = []
                engines = []
                for r in range(world):
                    e = Engine(64, C, K, k, proximal=_proximal(64, C, perm), boosting=B.ExponentialBoosting(C, k), distal=distal(C, K, tmp),
                               seed=3, shard_rank=r, shard_world=world, stream=None if r == 0 else engines[0].stream_handle() or "default")
                    engines.append(e)
                    e.import_tm_state(st)
                    parts.append(e.export_tm_state())
                same_state(st, merge_shard_states(parts, C, K), f"{world} shards K={K}")
        else:
            try:
                Engine(64, C, K, k, proximal=_proximal(64, C, np.zeros((C, 64))), boosting=B.ExponentialBoosting(C, k), distal=distal(C, K, tmp), shard_rank=0, shard_world=2)
                raise AssertionError("a sharded handle with 40 cells per column was accepted")
            except HtmError as e:
                assert "cell_dim" in str(e)
    # ---- a group of column shards in one process: the in-library loop in every plan (graphs of whole steps, eager, stepwise)
    from bithtm_amd.distributed import LocalGroup
    for world in (2, 8):
        group = LocalGroup(world, 64, 1024, 8, permanence=np.zeros((1024, 64)))
        group.upload_bank(rng.rand(5, 64) < 0.1)
        for steps, kw in ((1, {}), (2, {}), (37, {}), (9, dict(use_graph=False)), (6, dict(pipeline=False)), (3, dict(stepwise=True))):
            group.run(steps, **kw)
        group.process(rng.rand(64) < 0.1)
        for e in group.engines:
            e.check_capacity()
        group.export_tm_state()
    # ---- pre-populated pool, argument errors
    eng = Engine(0, 256, 16, 6, distal=B.PredictiveProjection(256 * 16, segment_capacity=256 * 16 * 3, segment_slots=64))
    eng.populate(3, synapses=20)
    try:
        eng.populate(3, synapses=20)
        raise AssertionError("populate on a used handle was accepted")
    except HtmError:
        pass
    try:
        eng.tm_step(np.array([1, 1]))
        raise AssertionError("a duplicate column was accepted")
    except HtmError:
        pass
    import ctypes
    counts = []
    for name in ("bithtm_stub_kernel_launches", "bithtm_stub_graph_launches"):
        fn = getattr(lib, name)
        fn.restype = ctypes.c_long
        counts.append(fn())
    assert counts[0] > 500 and counts[1] > 5, counts
    print(f"host sanitizer driver: ok ({counts[0]} kernel launches checked and skipped, {counts[1]} graph replays)")


def _proximal(I, C, perm):
    p = B.DenseProjection.__new__(B.DenseProjection)
    p.input_dim, p.output_dim = I, C
    p.permanence_threshold, p.permanence_increment, p.permanence_decrement = 0.0, 0.03, 0.015
    p._engine, p._permanence = None, perm
    return p


if __name__ == "__main__":
    main()
