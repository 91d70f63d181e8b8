"""NumPy restatement of the bitHTM timestep (Spatial Pooler + Temporal Memory).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the reference
lines it restates; paths are relative to /root/reference.  The restatement is written in
the *pull* form the GPU uses (one fixed-width row of presynaptic ids per segment) instead
of the reference's mirrored push/pull store, and it replaces the three places where the
reference's result is implementation-defined by the documented policies:

  * top-k ties (`regularizations.py:29`, np.argpartition)  -> (value desc, index asc),
    winners returned in ascending column order;
  * float32 exp (`regularizations.py:16`)                   -> oracle.fexp.exp_f32;
  * global np.random stream (`networks.py:87`, `projections.py:120,235`)
                                                            -> oracle.keyed_rng.draw24.

With those three substitutions applied to the reference through its own hooks
(oracle/ref_hooks.py) the two implementations agree bit-for-bit on every per-step output
and on the synapse store, including segment ids (tests/test_oracle_vs_reference.py).
"""

from dataclasses import dataclass, field
from types import SimpleNamespace

import numpy as np

from .fexp import exp_f32
from .keyed_rng import (draw24, draw32, draw_unit, STREAM_LEAST_USED, STREAM_GROWTH,
                        STREAM_SEGMENT_JITTER, STREAM_POPULATE_CELL, STREAM_POPULATE_PERM)

EPS32 = np.float32(1e-8)     # `epsilon=1e-8` compared against float32 arrays (weak scalar)


# --------------------------------------------------------------------------- parameters

@dataclass
class SPParams:
    """Defaults of DenseProjection (projections.py:7-11) + ExponentialBoosting
    (regularizations.py:5-8)."""
    permanence_mean: float = 0.0
    permanence_std: float = 0.1
    permanence_threshold: float = 0.0
    permanence_increment: float = 0.03
    permanence_decrement: float = 0.015
    boost_intensity: float = 0.3
    boost_momentum: float = 0.99


@dataclass
class TMParams:
    """Defaults of PredictiveProjection (projections.py:205-210)."""
    permanence_initial: float = 0.21
    permanence_threshold: float = 0.5
    permanence_increment: float = 0.1
    permanence_decrement: float = 0.1
    permanence_punishment: float = 0.01
    segment_activation_threshold: int = 15
    segment_matching_threshold: int = 15
    segment_sampling_synapses: int = 32


def sp_derived(p: SPParams, column_dim, active_columns):
    """The scalars the reference forms implicitly, with the dtype each one ends up in.

    projections.py:24  `input * (inc + dec) - dec`  (bool * float -> float64)
    regularizations.py:9,16,20,21  density, float32 coefficient / momentum / increment
    (Python floats are weak scalars against float32 arrays).
    """
    both = p.permanence_increment + p.permanence_decrement
    density = active_columns / column_dim
    return SimpleNamespace(
        delta_on=1.0 * both - p.permanence_decrement,
        delta_off=0.0 * both - p.permanence_decrement,
        coef32=np.float32(-(p.boost_intensity / density)),
        momentum32=np.float32(p.boost_momentum),
        increment32=np.float32(1.0 - p.boost_momentum),
    )


def tm_derived(p: TMParams):
    """projections.py:102 `act * (a - b) + b` for the (learn, punish) calls of :284-293."""
    a, b = p.permanence_increment, -p.permanence_decrement
    pa, pb = -p.permanence_punishment, 0.0
    return SimpleNamespace(
        learn_active=1.0 * (a - b) + b, learn_inactive=0.0 * (a - b) + b,
        learn_prune=min(a, b) < 0,
        punish_active=1.0 * (pa - pb) + pb, punish_inactive=0.0 * (pa - pb) + pb,
        punish_prune=min(pa, pb) < 0,
        initial32=np.float32(p.permanence_initial),
        threshold32=np.float32(p.permanence_threshold),
    )


# --------------------------------------------------------------------------- SP

def stable_topk(values, k):
    """Top-k policy replacing np.argpartition (regularizations.py:29): larger value first,
    lower index first among equals; result sorted by ascending index."""
    order = np.argsort(-np.asarray(values, dtype=np.float64), kind="stable")[:k]
    return np.sort(order).astype(np.int64)


def topk_is_unambiguous(values, k):
    """True when no tie straddles the k-th / (k+1)-th position."""
    v = np.sort(np.asarray(values, dtype=np.float64))[::-1]
    return k >= len(v) or v[k - 1] != v[k]


class SpatialPoolerOracle:
    """networks.py:7-35 with DenseProjection (projections.py:6-24), ExponentialBoosting
    (regularizations.py:4-21) and the stable top-k policy."""

    def __init__(self, input_dim, column_dim, active_columns, params=None, permanence=None):
        self.input_dim, self.column_dim, self.active_columns = input_dim, column_dim, active_columns
        self.params = params or SPParams()
        if permanence is None:   # projections.py:16, same expression => same global-RNG use
            permanence = (np.random.randn(column_dim, input_dim) * self.params.permanence_std
                          + self.params.permanence_mean)
        self.permanence = np.array(permanence, dtype=np.float64)
        assert self.permanence.shape == (column_dim, input_dim)
        self.duty_cycle = np.zeros(column_dim, dtype=np.float32)     # regularizations.py:13
        self.d = sp_derived(self.params, column_dim, active_columns)

    def overlaps(self, input_bits):
        """projections.py:18-21."""
        connected = self.permanence >= self.params.permanence_threshold
        return (connected & input_bits).sum(axis=1)

    def boost(self, overlaps):
        """regularizations.py:15-17: float32 factor, float64 product."""
        factor = exp_f32(self.d.coef32 * self.duty_cycle)
        return factor.astype(np.float64) * overlaps

    def step(self, input_bits, learning=True):
        """networks.py:26-35."""
        input_bits = np.asarray(input_bits, dtype=np.bool_)
        overlaps = self.overlaps(input_bits)
        boosted = self.boost(overlaps)
        active = stable_topk(boosted, self.active_columns)
        if learning:                                             # projections.py:23-24
            self.permanence[active] += np.where(input_bits, self.d.delta_on, self.d.delta_off)
        self.duty_cycle *= self.d.momentum32                     # regularizations.py:20
        self.duty_cycle[active] += self.d.increment32            # regularizations.py:21
        return SimpleNamespace(active_column=active, overlaps=overlaps, boosted_overlaps=boosted)


# --------------------------------------------------------------------------- TM

class TemporalMemoryOracle:
    """networks.py:38-128 + PredictiveProjection / SparseProjection (projections.py:27-293).

    Store (one row per segment, `slots` columns, grown by doubling like utils.py:99-102):
      seg_cell[s]   owning cell (segment_bundle, projections.py:226)
      seg_nsyn[s]   valid synapses (output_edges, projections.py:42)
      presyn[s, e]  presynaptic flat cell id or -1 (target half of output_edge, :63-64)
      perm[s, e]    float32 permanence (output_permanence, :44); -1.0 when never used
      segcount[n]   segments per cell (bundle_segments, :227)
    """

    def __init__(self, column_dim, cell_dim, params=None, seed=0, slots=32):
        self.column_dim, self.cell_dim = column_dim, cell_dim
        self.N = column_dim * cell_dim
        self.params = params or TMParams()
        assert self.params.segment_activation_threshold >= self.params.segment_matching_threshold
        self.d = tm_derived(self.params)
        self.seed = seed
        self.slots = slots
        self.S = 0
        self.seg_cell = np.zeros(0, dtype=np.int32)
        self.seg_nsyn = np.zeros(0, dtype=np.int32)
        self.presyn = np.full((0, slots), -1, dtype=np.int32)
        self.perm = np.full((0, slots), -1.0, dtype=np.float32)
        self.segcount = np.zeros(self.N, dtype=np.int32)
        self.step_index = 0
        self.eps = EPS32                       # TemporalMemory.process(epsilon=) (networks.py:91), as float32; 0 < eps <= 1
        # previous-step state (networks.py:57-65)
        self.prev_prediction = np.zeros((column_dim, cell_dim), dtype=np.bool_)
        self.prev_activation = np.zeros((column_dim, cell_dim), dtype=np.bool_)
        self.prev_winner = np.zeros(0, dtype=np.int64)     # empty, not None (networks.py:61)
        self.prev_distal = None

    # ---- store helpers
    def _ensure_rows(self, rows):
        cap = len(self.seg_cell)
        if rows <= cap:
            return
        new_cap = max(1, cap)
        while new_cap < rows:
            new_cap *= 2
        grow = new_cap - cap
        self.seg_cell = np.concatenate([self.seg_cell, np.full(grow, -1, np.int32)])
        self.seg_nsyn = np.concatenate([self.seg_nsyn, np.zeros(grow, np.int32)])
        self.presyn = np.concatenate([self.presyn, np.full((grow, self.slots), -1, np.int32)])
        self.perm = np.concatenate([self.perm, np.full((grow, self.slots), -1.0, np.float32)])

    def _ensure_slots(self, slots):
        if slots <= self.slots:
            return
        new_slots = self.slots
        while new_slots < slots:
            new_slots *= 2
        grow = new_slots - self.slots
        rows = len(self.seg_cell)
        self.presyn = np.concatenate([self.presyn, np.full((rows, grow), -1, np.int32)], axis=1)
        self.perm = np.concatenate([self.perm, np.full((rows, grow), -1.0, np.float32)], axis=1)
        self.slots = new_slots

    def _padded(self, activation):
        return np.concatenate([activation.reshape(-1), np.zeros(1, np.bool_)])

    def _targets(self, segs):
        ps = self.presyn[segs]
        valid = ps >= 0
        return ps, valid, np.where(valid, ps, self.N)

    # ---- synthetic pre-populated pool (twin of htm_populate / k_tm_populate; BASELINE.json configs[4])
    def populate(self, segments_per_cell, synapses=32, perm_lo=0.3, perm_hi=0.7, seed=0, cell_begin=0, cell_end=None):
        """Every cell in [cell_begin, cell_end) gets `segments_per_cell` segments of `synapses` synapses to keyed-random
        presynaptic cells, distinct within a segment (in synapse order, a cell already taken moves on to the next
        free cell id, mod N), permanences keyed-uniform in [perm_lo, perm_hi).  Segment ids are cell-major.  The
        reference has no counterpart (it grows its store step by step, projections.py:226); this is the scan
        stress SURVEY section 8(d) describes."""
        assert self.S == 0 and self.step_index == 0
        N = self.N
        cell_end = N if cell_end is None else cell_end
        spc, n = int(segments_per_cell), int(synapses)
        S = (cell_end - cell_begin) * spc
        gid = np.arange(S, dtype=np.int64)
        self._ensure_slots(n)
        self.seg_cell = (cell_begin + gid // spc).astype(np.int32)
        self.seg_nsyn = np.full(S, n, dtype=np.int32)
        self.presyn = np.full((S, self.slots), -1, dtype=np.int32)
        self.perm = np.full((S, self.slots), -1.0, dtype=np.float32)
        for lo in range(0, S, 1 << 18):                 # in slices: the intermediates stay small
            g = gid[lo:lo + (1 << 18)]
            cells, perms = populated_rows(N, g, n, perm_lo, perm_hi, seed)
            self.presyn[lo:lo + len(g), :n] = cells
            self.perm[lo:lo + len(g), :n] = perms
        self.segcount[cell_begin:cell_end] = spc
        self.S = S

    # ---- learning pieces
    def _update_permanence(self, segs, act_pad, d_active, d_inactive, prune):
        """projections.py:97-109: float64 sum, float32 store, prune on the float64 value."""
        if len(segs) == 0:
            return
        ps, valid, idx = self._targets(segs)
        change = np.where(act_pad[idx], d_active, d_inactive)
        updated = self.perm[segs].astype(np.float64) + valid * change
        self.perm[segs] = updated.astype(np.float32)
        if prune:
            negative = updated < 0.0
            self.seg_nsyn[segs] -= (valid & negative).sum(axis=1).astype(np.int32)
            self.presyn[segs] = np.where(negative, -1, ps)

    def _grow(self, segs, act_pad, winners, step, chunk=256):
        """projections.py:111-161: each learning segment samples, without replacement and by
        ascending keyed priority, up to `sampling - active synapses` previous winner cells it
        is not yet connected to; new synapses start at permanence_initial."""
        n_w = len(winners)
        sample = self.params.segment_sampling_synapses
        if len(segs) == 0 or n_w == 0:
            return
        position = np.full(self.N + 1, -1, dtype=np.int64)
        position[winners] = np.arange(n_w)
        col_index = np.arange(n_w, dtype=np.int64)
        for lo in range(0, len(segs), chunk):
            part = segs[lo:lo + chunk]
            ps, valid, idx = self._targets(part)
            active_valid = act_pad[idx].sum(axis=1)
            n_add = np.clip(sample - active_valid, 0, min(sample, n_w))
            key = draw24(self.seed, STREAM_GROWTH, step, part[:, None], winners[None, :]).astype(np.int64)
            key = key * n_w + col_index                 # ties: lower winner index first
            pos = position[idx]
            rows, cols = np.nonzero(pos >= 0)
            key[rows, pos[rows, cols]] = np.iinfo(np.int64).max      # already connected
            absent = n_w - np.bincount(rows, minlength=len(part))
            take = np.minimum(n_add, absent)
            kmax = int(take.max(initial=0))
            if kmax == 0:
                continue
            if kmax < n_w:
                smallest = np.argpartition(key, kmax - 1, axis=1)[:, :kmax]
            else:
                smallest = np.broadcast_to(col_index, key.shape).copy()
            smallest_key = np.take_along_axis(key, smallest, axis=1)
            order = np.argsort(smallest_key, axis=1)
            smallest = np.take_along_axis(smallest, order, axis=1)
            for i in np.flatnonzero(take > 0):
                seg, t = part[i], int(take[i])
                cells = np.sort(winners[smallest[i, :t]])
                free = np.flatnonzero(self.presyn[seg] < 0)
                if len(free) < t:
                    self._ensure_slots(self.slots + (t - len(free)))
                    free = np.flatnonzero(self.presyn[seg] < 0)
                self.presyn[seg, free[:t]] = cells
                self.perm[seg, free[:t]] = self.d.initial32
                self.seg_nsyn[seg] += t

    def _learn(self, winner_flat, active_column, step):
        """projections.py:257-293 (PredictiveProjection.update) + add_output (:79-95)."""
        d, p, K = self.prev_distal, self.params, self.cell_dim
        if d is None:                                            # projections.py:258-259
            return
        m = d.matching_segment
        mcell = self.seg_cell[m]
        is_winner = np.zeros(self.N, dtype=np.bool_)
        is_winner[winner_flat] = True
        unpredicted = d.prediction[mcell] < 1e-8                              # :266
        best = np.abs(d.matching_segment_jittered_potential - d.max_jittered_potential[mcell]) < self.eps  # :267
        learning = m[is_winner[mcell] & (d.matching_segment_active | (unpredicted & best))]     # :268
        column_active = np.zeros(self.column_dim, dtype=np.bool_)
        column_active[active_column] = True
        punished = m[~column_active[mcell // K]]                              # :269, networks.py:107-111

        unaccounted = winner_flat[d.max_jittered_potential[winner_flat] < self.eps]   # :271
        if len(unaccounted):
            recycled = np.flatnonzero(self.seg_nsyn[:self.S] < p.segment_matching_threshold)[:len(unaccounted)]  # :80-81
            n_r = len(recycled)
            np.subtract.at(self.segcount, self.seg_cell[recycled], 1)         # :275-276
            self.presyn[recycled] = -1                                        # :82-85
            self.perm[recycled] = -1.0
            self.seg_nsyn[recycled] = 0
            self.segcount[unaccounted] += 1                                   # :277
            self.seg_cell[recycled] = unaccounted[:n_r]                       # :278
            n_new = len(unaccounted) - n_r
            fresh = np.arange(self.S, self.S + n_new)
            if n_new:                                                         # :90-94, :279-280
                self._ensure_rows(self.S + n_new)
                self.seg_cell[fresh] = unaccounted[n_r:]
                self.seg_nsyn[fresh] = 0
                self.presyn[fresh] = -1
                self.perm[fresh] = -1.0
                self.S += n_new
            learning = np.concatenate([learning, recycled, fresh])            # :281

        act_pad = self._padded(self.prev_activation)                          # :283
        self._update_permanence(learning, act_pad, self.d.learn_active, self.d.learn_inactive, self.d.learn_prune)
        if self.prev_winner is not None:                                      # :191-192
            self._grow(learning.astype(np.int64), act_pad, self.prev_winner, step)
        self._update_permanence(punished, act_pad, self.d.punish_active, self.d.punish_inactive, self.d.punish_prune)

    # ---- scan
    def _scan(self, activation, step):
        """projections.py:245-255 + fill_jittered_potential_info (:229-239)."""
        p = self.params
        act_pad = self._padded(activation)
        segs = np.arange(self.S)
        _, _, idx = self._targets(segs)
        hit = act_pad[idx]
        potential = hit.sum(axis=1).astype(np.int64)
        matching = np.flatnonzero(potential >= p.segment_matching_threshold)
        connected = (self.perm[matching] >= self.d.threshold32) & hit[matching]
        activation_count = connected.sum(axis=1).astype(np.int64)
        active = activation_count >= p.segment_activation_threshold
        cells = self.seg_cell[matching]
        prediction = np.bincount(cells, weights=active, minlength=self.N).astype(np.float64)
        u = draw_unit(self.seed, STREAM_SEGMENT_JITTER, step, matching)
        jittered = (potential[matching].astype(np.float32).astype(np.float64) + u).astype(np.float32)
        cell_max = np.zeros(self.N, dtype=np.float32)
        np.maximum.at(cell_max, cells, jittered)
        return SimpleNamespace(
            prediction=prediction, segment_potential=potential, matching_segment=matching,
            matching_segment_activation=activation_count, matching_segment_active=active,
            max_jittered_potential=cell_max, matching_segment_jittered_potential=jittered)

    # ---- one timestep
    def step(self, active_column, learning=True, return_winner_cell=True, prev_state=None):
        """networks.py:91-128.  `active_column` must be sorted ascending (stable top-k).  `prev_state`: a state this
        method returned earlier, to be used instead of the last one (networks.py:92-93)."""
        C, K = self.column_dim, self.cell_dim
        if prev_state is not None:
            self.prev_prediction, self.prev_activation = prev_state.cell_prediction, prev_state.cell_activation
            self.prev_winner = None if prev_state.winner_cell is None else prev_state.winner_cell[0].astype(np.int64) * K + prev_state.winner_cell[1]
            self.prev_distal = prev_state.distal_state
        active_column = np.asarray(active_column, dtype=np.int64)
        t = self.step_index
        predicted = self.prev_prediction[active_column]                       # :96
        bursting = ~predicted.any(axis=1)                                     # :97
        flat = active_column[:, None] * K + np.arange(K)
        winner_flat = None
        if learning or return_winner_cell:
            if self.prev_distal is None:                                      # :74-75
                column_matching = np.zeros(len(active_column), dtype=np.bool_)
                best = np.zeros((len(active_column), K), dtype=np.bool_)
            else:                                                             # :76-81
                cell_max = self.prev_distal.max_jittered_potential.reshape(C, K)[active_column]
                column_max = cell_max.max(axis=1, keepdims=True) if len(active_column) else cell_max[:, :1]
                column_matching = (column_max >= self.params.segment_matching_threshold)[:, 0]
                best = np.abs(cell_max - column_max) < self.eps
            count = self.segcount.reshape(C, K)[active_column].astype(np.float32)     # :85-86
            u = draw_unit(self.seed, STREAM_LEAST_USED, t, flat)
            jittered = (count.astype(np.float64) + u).astype(np.float32)              # :87
            least = (np.abs(jittered - jittered.min(axis=1, keepdims=True)) < self.eps
                     if len(active_column) else jittered.astype(np.bool_))            # :88
            winner = predicted | (bursting[:, None] & np.where(column_matching[:, None], best, least))  # :102
            winner_flat = flat[winner]                                        # :103-104 (row-major)
        if learning:
            self._learn(winner_flat, active_column, t)                        # :106-113
        activated = predicted | bursting[:, None]                             # :115
        activation = np.zeros((C, K), dtype=np.bool_)
        activation[active_column] = activated                                 # :118-119
        distal = self._scan(activation, t)                                    # :121
        prediction = distal.prediction.reshape(C, K) > 1e-8                   # :122

        self.prev_prediction, self.prev_activation = prediction, activation
        self.prev_winner = winner_flat
        self.prev_distal = distal
        self.step_index += 1
        active_flat = flat[activated]
        return SimpleNamespace(
            active_cell=(active_flat // K, active_flat % K),
            winner_cell=None if winner_flat is None else (winner_flat // K, winner_flat % K),
            cell_activation=activation, cell_prediction=prediction,
            active_column_bursting=bursting[:, None], distal_state=distal)

    # ---- state hand-off (same arrays the HIP library exports / imports)
    def export_state(self):
        S = self.S
        d = self.prev_distal
        out = dict(
            S=np.int64(S), slots=np.int64(self.slots), step_index=np.int64(self.step_index),
            seg_cell=self.seg_cell[:S].copy(), seg_nsyn=self.seg_nsyn[:S].copy(),
            presyn=self.presyn[:S].copy(), perm=self.perm[:S].copy(),
            segcount=self.segcount.copy(),
            prev_prediction=self.prev_prediction.copy(), prev_activation=self.prev_activation.copy(),
            prev_winner=(np.zeros(0, np.int64) if self.prev_winner is None else self.prev_winner.copy()),
            has_prev_winner=np.bool_(self.prev_winner is not None),
            has_distal=np.bool_(d is not None),
        )
        if d is not None:
            out.update(
                segment_potential=d.segment_potential.copy(), matching_segment=d.matching_segment.copy(),
                matching_segment_activation=d.matching_segment_activation.copy(),
                matching_segment_active=d.matching_segment_active.copy(),
                matching_segment_jittered_potential=d.matching_segment_jittered_potential.copy(),
                max_jittered_potential=d.max_jittered_potential.copy(), prediction=d.prediction.copy())
        return out

    def import_state(self, st):
        S, slots = int(st["S"]), int(st["slots"])
        self.slots = slots
        self.S = S
        self.seg_cell = np.array(st["seg_cell"], dtype=np.int32)
        self.seg_nsyn = np.array(st["seg_nsyn"], dtype=np.int32)
        self.presyn = np.array(st["presyn"], dtype=np.int32).reshape(S, slots)
        self.perm = np.array(st["perm"], dtype=np.float32).reshape(S, slots)
        self.segcount = np.array(st["segcount"], dtype=np.int32)
        self.step_index = int(st["step_index"])
        self.prev_prediction = np.array(st["prev_prediction"], dtype=np.bool_).reshape(self.column_dim, self.cell_dim)
        self.prev_activation = np.array(st["prev_activation"], dtype=np.bool_).reshape(self.column_dim, self.cell_dim)
        self.prev_winner = np.array(st["prev_winner"], dtype=np.int64) if bool(st["has_prev_winner"]) else None
        if bool(st["has_distal"]):
            self.prev_distal = SimpleNamespace(
                segment_potential=np.array(st["segment_potential"], dtype=np.int64),
                matching_segment=np.array(st["matching_segment"], dtype=np.int64),
                matching_segment_activation=np.array(st["matching_segment_activation"], dtype=np.int64),
                matching_segment_active=np.array(st["matching_segment_active"], dtype=np.bool_),
                matching_segment_jittered_potential=np.array(st["matching_segment_jittered_potential"], dtype=np.float32),
                max_jittered_potential=np.array(st["max_jittered_potential"], dtype=np.float32),
                prediction=np.array(st["prediction"], dtype=np.float64))
        else:
            self.prev_distal = None


def populated_rows(N, gid, synapses, perm_lo, perm_hi, seed):
    """The synapses htm_populate / TemporalMemoryOracle.populate give the segments with ids `gid` (any subset, any order):
    presynaptic flat cell ids int64[len(gid), synapses] and float32 permanences -- keyed by the segment id alone, so a
    sample of a pool too large for the host can be regenerated and checked (bench.py's configs[4] leg)."""
    g = np.asarray(gid, dtype=np.int64)
    n = int(synapses)
    idx = np.arange(n)
    cells = ((draw32(seed, STREAM_POPULATE_CELL, 0, g[:, None], idx[None, :]).astype(np.uint64) * np.uint64(N))
             >> np.uint64(32)).astype(np.int64)
    srt = np.sort(cells, axis=1)
    for r in np.flatnonzero((srt[:, 1:] == srt[:, :-1]).any(axis=1)):      # rare: settle in synapse order
        row = cells[r]
        for i in range(1, n):
            while row[i] in row[:i]:
                row[i] = (row[i] + 1) % N
    u = draw24(seed, STREAM_POPULATE_PERM, 0, g[:, None], idx[None, :]).astype(np.float64) * (1.0 / 16777216.0)
    return cells, (perm_lo + (perm_hi - perm_lo) * u).astype(np.float32)


def canonical_synapses(seg_cell, presyn, perm):
    """Slot-order-free view of a synapse store: for every segment id, its owning cell and
    its valid synapses as (presynaptic id, permanence) sorted by presynaptic id."""
    out = []
    for s in range(len(seg_cell)):
        valid = presyn[s] >= 0
        ids = presyn[s][valid]
        order = np.argsort(ids, kind="stable")
        out.append((int(seg_cell[s]), ids[order].astype(np.int64), perm[s][valid][order].astype(np.float32)))
    return out


class HTMOracle:
    """networks.py:131-149."""

    def __init__(self, input_dim, column_dim, cell_dim, active_columns=None, seed=0,
                 sp_params=None, tm_params=None, permanence=None):
        if active_columns is None:
            active_columns = round(column_dim * 0.02)                         # networks.py:137
        self.input_dim, self.column_dim, self.cell_dim = input_dim, column_dim, cell_dim
        self.active_columns = active_columns
        self.spatial_pooler = SpatialPoolerOracle(input_dim, column_dim, active_columns, sp_params, permanence)
        self.temporal_memory = TemporalMemoryOracle(column_dim, cell_dim, tm_params, seed)

    def step(self, input_bits, learning=True):
        sp_state = self.spatial_pooler.step(input_bits, learning=learning)
        tm_state = self.temporal_memory.step(sp_state.active_column, learning=learning)
        return sp_state, tm_state
