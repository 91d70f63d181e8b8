"""Replay a golden trajectory (tests/golden/traj_*.npz) through any implementation.

An implementation is wrapped as an object with

    step(input_bits: bool[I], learning: bool) -> dict     per-step outputs (see FIELDS)
    store() -> dict                                       synapse store + SP state

and is compared field by field, bit-exactly, with what the unmodified reference produced.
Used for the oracle (CPU) and for the HIP path (GPU).  Not collected by pytest.
"""

import zlib

import numpy as np

from oracle import SPParams, TMParams, canonical_synapses

FIELDS = ("active_column", "overlaps", "boosted", "bursting", "act_bits", "pred_bits", "winner",
          "matching", "match_pot", "match_act", "match_active", "S")


def load(path):
    z = np.load(path)
    g = {k: z[k] for k in z.files}
    g["sp_params"] = _params(g, "sp_", SPParams)
    g["tm_params"] = _params(g, "tm_", TMParams)
    return g


def _params(g, prefix, cls):
    names = [f for f in cls.__dataclass_fields__ if prefix + f in g]
    if not names:
        return None
    return cls(**{f: cls.__dataclass_fields__[f].type(g[prefix + f]) for f in names})


def initial_permanence(g):
    """projections.py:16 under np.random.seed(seed); verified against the stored digest."""
    p = g["sp_params"] or SPParams()
    np.random.seed(int(g["seed"]))
    perm = np.random.randn(int(g["column_dim"]), int(g["input_dim"])) * p.permanence_std + p.permanence_mean
    crc = np.array([zlib.crc32(perm[i].tobytes()) for i in range(perm.shape[0])], dtype=np.uint32)
    assert np.array_equal(crc, g["sp_perm0_crc"]), "regenerated initial SP permanence differs from the fixture"
    return perm


def unpack_inputs(g):
    I = int(g["input_dim"])
    return np.unpackbits(g["inputs"], axis=1, bitorder="little")[:, :I].astype(np.bool_)


def _rag(g, key, t):
    off = g[key + "_off"]
    return g[key][off[t]:off[t + 1]]


def check_step(g, t, out, boosted_index):
    def eq(name, got, want):
        got, want = np.asarray(got), np.asarray(want)
        assert got.shape == want.shape, f"step {t}: {name}: shape {got.shape} != {want.shape}"
        if not np.array_equal(got, want):
            bad = np.flatnonzero((got != want).reshape(-1))
            raise AssertionError(f"step {t}: {name}: {len(bad)} mismatches, first at {bad[:6]}: "
                                 f"got {got.reshape(-1)[bad[:6]]} want {want.reshape(-1)[bad[:6]]}")

    eq("active_column", out["active_column"], g["active_column"][t])
    assert zlib.crc32(np.asarray(out["overlaps"]).astype(np.int32).tobytes()) == int(g["overlaps_crc"][t]), \
        f"step {t}: overlaps crc"
    if t in boosted_index:
        j = boosted_index[t]
        eq("overlaps", np.asarray(out["overlaps"]).astype(np.int16), g["overlaps"][j])
        eq("boosted bits", np.asarray(out["boosted"], dtype=np.float64).view(np.int64), g["boosted"][j].view(np.int64))
    eq("bursting", out["bursting"], g["bursting"][t])
    eq("act_bits", out["act_bits"], g["act_bits"][t])
    eq("pred_bits", out["pred_bits"], g["pred_bits"][t])
    eq("winner", out["winner"], _rag(g, "winner", t))
    eq("S", int(out["S"]), int(g["S"][t]))
    eq("matching", out["matching"], _rag(g, "matching", t))
    eq("match_pot", out["match_pot"], _rag(g, "match_pot", t))
    eq("match_act", out["match_act"], _rag(g, "match_act", t))
    eq("match_active", out["match_active"], _rag(g, "match_active", t))


def check_store(g, t, st):
    pre = f"ck{t}_"
    S = len(g[pre + "seg_cell"])
    assert int(st["S"]) == S, f"checkpoint {t}: S {st['S']} != {S}"
    assert np.array_equal(st["seg_cell"][:S], g[pre + "seg_cell"]), f"checkpoint {t}: seg_cell"
    assert np.array_equal(st["seg_nsyn"][:S], g[pre + "seg_nsyn"]), f"checkpoint {t}: seg_nsyn"
    assert np.array_equal(st["segcount"], g[pre + "segcount"]), f"checkpoint {t}: segcount"
    canon = canonical_synapses(st["seg_cell"][:S], st["presyn"][:S], st["perm"][:S])
    off = g[pre + "syn_off"]
    for s, (_, ids, perms) in enumerate(canon):
        want_ids = g[pre + "syn_presyn"][off[s]:off[s + 1]]
        want_perm = g[pre + "syn_perm"][off[s]:off[s + 1]]
        assert np.array_equal(ids, want_ids), f"checkpoint {t}: segment {s} presynaptic ids"
        assert np.array_equal(perms.view(np.int32), want_perm.view(np.int32)), f"checkpoint {t}: segment {s} permanence bits"
    assert np.array_equal(np.asarray(st["duty"], dtype=np.float32).view(np.int32), g[pre + "duty"].view(np.int32)), \
        f"checkpoint {t}: duty cycle bits"
    perm = np.ascontiguousarray(st["sp_permanence"], dtype=np.float64)
    crc = np.array([zlib.crc32(perm[i].tobytes()) for i in range(perm.shape[0])], dtype=np.uint32)
    bad = np.flatnonzero(crc != g[pre + "sp_perm_crc"])
    assert len(bad) == 0, f"checkpoint {t}: SP permanence rows differ: {bad[:8]}"
    if "segment_potential" in st:
        assert np.array_equal(np.asarray(st["segment_potential"])[:S].astype(np.int16), g[pre + "segment_potential"]), \
            f"checkpoint {t}: segment_potential"
    if "max_jittered_potential" in st:
        assert np.array_equal(np.asarray(st["max_jittered_potential"], dtype=np.float32).view(np.int32),
                              g[pre + "max_jittered_potential"].view(np.int32)), f"checkpoint {t}: max_jittered_potential"


def replay(g, impl, steps=None):
    xs = unpack_inputs(g)
    n = int(g["steps"]) if steps is None else min(int(steps), int(g["steps"]))
    boosted_index = {int(s): j for j, s in enumerate(g["boosted_steps"])}
    checkpoints = set(int(c) for c in g["checkpoints"])
    for t in range(n):
        out = impl.step(xs[t], bool(g["learning"][t]))
        check_step(g, t, out, boosted_index)
        if t in checkpoints:
            check_store(g, t, impl.store())
    return n


class OracleImpl:
    """Adapter: oracle.HTMOracle -> replay interface."""

    def __init__(self, g):
        from oracle import HTMOracle
        self.K = int(g["cell_dim"])
        self.o = HTMOracle(int(g["input_dim"]), int(g["column_dim"]), self.K,
                           active_columns=int(g["active_columns"]), seed=int(g["seed"]),
                           sp_params=g["sp_params"], tm_params=g["tm_params"],
                           permanence=initial_permanence(g))

    def step(self, x, learning):
        sp, tm = self.o.step(x, learning=learning)
        d = tm.distal_state
        return dict(
            active_column=sp.active_column, overlaps=sp.overlaps, boosted=sp.boosted_overlaps,
            bursting=tm.active_column_bursting[:, 0],
            act_bits=np.packbits(tm.cell_activation.reshape(-1), bitorder="little"),
            pred_bits=np.packbits(tm.cell_prediction.reshape(-1), bitorder="little"),
            winner=tm.winner_cell[0] * self.K + tm.winner_cell[1],
            matching=d.matching_segment, match_pot=d.segment_potential[d.matching_segment],
            match_act=d.matching_segment_activation, match_active=d.matching_segment_active,
            S=len(d.segment_potential))

    def store(self):
        tm, sp = self.o.temporal_memory, self.o.spatial_pooler
        S = tm.S
        return dict(S=S, seg_cell=tm.seg_cell[:S], seg_nsyn=tm.seg_nsyn[:S], presyn=tm.presyn[:S], perm=tm.perm[:S],
                    segcount=tm.segcount, duty=sp.duty_cycle, sp_permanence=sp.permanence,
                    segment_potential=tm.prev_distal.segment_potential,
                    max_jittered_potential=tm.prev_distal.max_jittered_potential)
