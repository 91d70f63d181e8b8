"""Column-sharded form of the oracle timestep.  TEST INFRASTRUCTURE ONLY.

This file pins the multi-GPU protocol the HIP engine implements ("Multi-GPU"):
rank r of R owns the contiguous columns [c0, c1) -- their Spatial Pooler rows and duty cycles, their cells, and
every distal segment whose owning cell is one of those cells.  Segment ids are GLOBAL (what an unsharded run would
assign: they key the random draws and order the recycling); every rank keeps, for the whole id space, only whether
an id has fewer synapses than the matching threshold (`dead`, what projections.py:80-81 looks at).

One exchange per timestep (an all-gather of fixed-size per-rank records):

    candidates[n]    a SUPERSET of the rank's own top-min(k, own columns) columns by (boosted desc, column asc) --
                     exactly that set (offer="exact"), or together with every other own column whose boosted overlap
                     shares the leading bits of the last one's (offer="bin": what the HIP engine's local select hands
                     over when the whole threshold bin of its histogram fits the record's CAP slots; the global
                     top-k does not care) --, in ascending
                     column order, each with its boosted overlap (f64), its bursting bit and the winner-cell and
                     needs-a-new-segment words the column WOULD have if it became active (computable before the
                     global top-k: they depend only on the rank's own previous predictions / segment maxima /
                     segment counts; the active word is all cells when bursting, the winner word otherwise)
    dead[]           ids of own segments that dropped below the matching threshold during the previous step's
                     learning (the lowest-id-first recycling rule, projections.py:80-81, is global)

After the gather every rank performs the identical global top-k over the candidates of all ranks and the identical
segment-id allocation, so no further communication is needed and the R-way result equals the 1-way result bit for
bit.  `all_gather` is injected: torch.distributed (gloo) in tests/test_sharded_gloo.py.
"""

from types import SimpleNamespace

import numpy as np

from .fexp import exp_f32
from .htm_oracle import (EPS32, SPParams, TMParams, TemporalMemoryOracle, sp_derived, stable_topk)
from .keyed_rng import draw_unit, STREAM_LEAST_USED, STREAM_SEGMENT_JITTER


def shard_range(rank, world, column_dim):
    """Contiguous, equal column shards (column_dim must divide evenly)."""
    if column_dim % world:
        raise ValueError("column_dim must be a multiple of the number of shards")
    per = column_dim // world
    return rank * per, (rank + 1) * per


def cand_cap(n_cand, n_local):
    """Candidate slots of a record (bithtm_amd/csrc/htm_sp_kernels.h: shard_cand_cap)."""
    return min(n_local, n_cand + max(64, n_cand // 4 if n_cand * 8 > n_local else n_cand // 2))


class ShardedHTMOracle:
    def __init__(self, rank, world, input_dim, column_dim, cell_dim, active_columns=None, seed=0,
                 sp_params=None, tm_params=None, permanence=None, offer="exact"):
        if active_columns is None:
            active_columns = round(column_dim * 0.02)
        self.rank, self.world = rank, world
        self.input_dim, self.column_dim, self.cell_dim, self.k = input_dim, column_dim, cell_dim, active_columns
        self.c0, self.c1 = shard_range(rank, world, column_dim)
        self.n_cand = min(self.k, self.c1 - self.c0)
        self.cap = cand_cap(self.n_cand, self.c1 - self.c0)
        assert offer in ("exact", "bin")
        self.offer = offer
        self.spp = sp_params or SPParams()
        self.d_sp = sp_derived(self.spp, column_dim, active_columns)
        assert permanence is not None and permanence.shape == (column_dim, input_dim)
        self.permanence = np.array(permanence[self.c0:self.c1], dtype=np.float64)      # own rows only
        self.duty = np.zeros(self.c1 - self.c0, dtype=np.float32)
        # the TM store is indexed by global id; a rank only ever touches the rows of segments it owns
        self.tm = TemporalMemoryOracle(column_dim, cell_dim, tm_params, seed)
        self.hot_budget = min(self.cap, 4096 // world)    # entries of a rank's hot list the global select looks at
        self.hot_target = max(1, min(self.n_cand, self.hot_budget // 2))
        self.hot_selects = 0                              # steps whose global top-k was settled among the hot lists
        self.dead = np.zeros(0, dtype=np.bool_)           # replicated: id has fewer synapses than the matching threshold
        self.dead_out = np.zeros(0, dtype=np.int64)       # reported with the next exchange

    # ---- ownership
    def owns_cell(self, flat):
        col = np.asarray(flat) // self.cell_dim
        return (col >= self.c0) & (col < self.c1)

    def owned_segments(self):
        return self.owns_cell(self.tm.seg_cell[:self.tm.S])

    def _ensure_dead(self, n):
        if len(self.dead) < n:
            self.dead = np.concatenate([self.dead, np.zeros(max(n, 2 * len(self.dead)) - len(self.dead), dtype=np.bool_)])

    # ---- phase A: everything that needs only this rank's own state
    def local_record(self, input_bits):
        tm, K = self.tm, self.cell_dim
        connected = self.permanence >= self.spp.permanence_threshold
        overlaps = (connected & input_bits).sum(axis=1)
        boosted = exp_f32(self.d_sp.coef32 * self.duty).astype(np.float64) * overlaps
        cand = stable_topk(boosted, self.n_cand)           # own candidates, ascending (local) column
        hot_slot, hot_floor = None, 0.0
        if self.offer == "bin" and len(cand):
            # everything down to the leading bits (sign, exponent, 7 mantissa bits) of the weakest candidate, if it fits
            lead = boosted.view(np.int64) >> 45
            wide = np.flatnonzero(lead >= lead[cand].min())
            if len(wide) <= self.cap:
                cand = wide
                # the hot list: the same cut at the hot_target-th largest key; its floor = the lowest value with those leading bits
                lead_hot = lead[stable_topk(boosted, self.hot_target)].min()
                hot = np.flatnonzero(lead >= lead_hot)
                if len(hot) <= self.hot_budget and self.cap <= 65536:
                    hot_slot = np.searchsorted(cand, hot)
                    hot_floor = float(np.array([lead_hot << 45], dtype=np.int64).view(np.float64)[0])
        cols = cand + self.c0
        predicted = tm.prev_prediction[cols]
        bursting = ~predicted.any(axis=1)
        flat = cols[:, None] * K + np.arange(K)
        if tm.prev_distal is None:
            column_matching = np.zeros(len(cols), dtype=np.bool_)
            best = np.zeros((len(cols), K), dtype=np.bool_)
            has_match = np.zeros((len(cols), K), dtype=np.bool_)
        else:
            cell_max = tm.prev_distal.max_jittered_potential.reshape(self.column_dim, K)[cols]
            column_max = cell_max.max(axis=1, keepdims=True)
            column_matching = (column_max >= tm.params.segment_matching_threshold)[:, 0]
            best = np.abs(cell_max - column_max) < EPS32
            has_match = ~(cell_max < EPS32)
        count = tm.segcount.reshape(self.column_dim, K)[cols].astype(np.float32)
        u = draw_unit(tm.seed, STREAM_LEAST_USED, tm.step_index, flat)
        jittered = (count.astype(np.float64) + u).astype(np.float32)
        least = np.abs(jittered - jittered.min(axis=1, keepdims=True)) < EPS32
        winner = predicted | (bursting[:, None] & np.where(column_matching[:, None], best, least))
        unacc = winner & ~has_match if tm.prev_distal is not None else np.zeros_like(winner)
        # (a rank that cut its candidates exactly sends no hot list, hot_slot = None, as the HIP engine does)
        return SimpleNamespace(overlaps=overlaps, boosted_all=boosted, boosted=boosted[cand], col=cols.astype(np.int64),
                               bursting=bursting, win=winner, unacc=unacc, dead=self.dead_out.copy(), hot_slot=hot_slot,
                               hot_floor=hot_floor)

    # ---- phase B: identical global decisions + this rank's share of the work
    def finish_step(self, input_bits, records, learning=True):
        tm, K, C = self.tm, self.cell_dim, self.column_dim
        p, d = tm.params, tm.d
        t = tm.step_index
        boosted = np.concatenate([r.boosted for r in records])       # R x KL candidates, ascending column overall
        col = np.concatenate([r.col for r in records])
        win_spec = np.concatenate([r.win for r in records])
        unacc_spec = np.concatenate([r.unacc for r in records])
        burst_spec = np.concatenate([r.bursting for r in records])
        assert np.all(np.diff(col) > 0)
        chosen = stable_topk(boosted, self.k)                           # value desc, then candidate order = column asc
        # the short way (what the HIP engine's global select does when it can): every rank sent a hot list, the lists hold k
        # keys between them and the k-th largest of those is at or above every list's floor (every candidate at or above a
        # rank's floor is in its list) -- then the top-k of the hot keys alone is the top-k of all
        if all(r.hot_slot is not None for r in records) and sum(len(r.hot_slot) for r in records) >= self.k:
            base = np.cumsum([0] + [len(r.boosted) for r in records[:-1]])
            hot = np.concatenate([b + r.hot_slot for b, r in zip(base, records)])
            top = stable_topk(boosted[hot], self.k)
            if boosted[hot][top].min() >= max(r.hot_floor for r in records):
                assert np.array_equal(hot[top], chosen), "hot lists select differently"
                self.hot_selects += 1
        active = col[chosen]                                            # identical everywhere, ascending
        # deaths every rank reported (this one's own included): recyclable from now on
        self._ensure_dead(tm.S)
        for rec in records:
            if len(rec.dead):
                self.dead[rec.dead] = True
        # SP learning and duty cycle on the own rows
        mine = active[(active >= self.c0) & (active < self.c1)] - self.c0
        if learning:
            self.permanence[mine] += np.where(input_bits, self.d_sp.delta_on, self.d_sp.delta_off)
        self.duty *= self.d_sp.momentum32
        self.duty[mine] += self.d_sp.increment32

        bursting = burst_spec[chosen]
        win = win_spec[chosen]
        act = np.where(bursting[:, None], True, win)                    # networks.py:115
        flat = active[:, None] * K + np.arange(K)
        winner_flat = flat[win]
        activation = np.zeros((C, K), dtype=np.bool_)
        activation[active] = act
        dead_out = []
        if learning and tm.prev_distal is not None:
            dd = tm.prev_distal
            m = dd.matching_segment                                            # owned segments only
            mcell = tm.seg_cell[m]
            is_winner = np.zeros(tm.N, dtype=np.bool_)
            is_winner[winner_flat] = True
            unpredicted = dd.prediction[mcell] < 1e-8
            best = np.abs(dd.matching_segment_jittered_potential - dd.max_jittered_potential[mcell]) < EPS32
            learning_seg = m[is_winner[mcell] & (dd.matching_segment_active | (unpredicted & best))]
            column_active = np.zeros(C, dtype=np.bool_)
            column_active[active] = True
            punished = m[~column_active[mcell // K]]
            # allocation: every rank takes the same decision (projections.py:79-95, 271-281)
            unaccounted = flat[unacc_spec[chosen]]
            n_w = -1 if tm.prev_winner is None else len(tm.prev_winner)
            if len(unaccounted):
                recycled = np.flatnonzero(self.dead[:tm.S])[:len(unaccounted)]
                n_r = len(recycled)
                n_new = len(unaccounted) - n_r
                fresh = np.arange(tm.S, tm.S + n_new)
                tm._ensure_rows(tm.S + n_new)
                self._ensure_dead(tm.S + n_new)
                old_cells = tm.seg_cell[recycled]
                own_old = self.owns_cell(old_cells)
                np.subtract.at(tm.segcount, old_cells[own_old], 1)
                ids = np.concatenate([recycled, fresh]).astype(np.int64)
                tm.seg_cell[ids] = unaccounted          # (the assignment is a replicated decision: every rank knows it)
                own_new = self.owns_cell(unaccounted)
                tm.segcount[unaccounted[own_new]] += 1
                tm.presyn[ids] = -1                     # rows given up / taken: empty either way
                tm.perm[ids] = -1.0
                tm.seg_nsyn[ids] = 0
                # what becomes of the row once its owner has grown it is known everywhere
                grown = min(p.segment_sampling_synapses, n_w) if n_w > 0 else 0
                self.dead[ids] = grown < p.segment_matching_threshold
                tm.S += n_new
                learning_seg = np.concatenate([learning_seg, ids[own_new]])
            act_pad = tm._padded(tm.prev_activation)
            thr = p.segment_matching_threshold
            before = tm.seg_nsyn[learning_seg].copy()
            tm._update_permanence(learning_seg, act_pad, d.learn_active, d.learn_inactive, d.learn_prune)
            if tm.prev_winner is not None:
                tm._grow(learning_seg.astype(np.int64), act_pad, tm.prev_winner, t)
            dead_out.append(learning_seg[(before >= thr) & (tm.seg_nsyn[learning_seg] < thr)])
            before = tm.seg_nsyn[punished].copy()
            tm._update_permanence(punished, act_pad, d.punish_active, d.punish_inactive, d.punish_prune)
            dead_out.append(punished[(before >= thr) & (tm.seg_nsyn[punished] < thr)])
        self.dead_out = np.concatenate(dead_out).astype(np.int64) if dead_out else np.zeros(0, dtype=np.int64)

        distal = self._scan_owned(activation, t)
        prediction = distal.prediction.reshape(C, K) > 1e-8
        tm.prev_prediction, tm.prev_activation = prediction, activation
        tm.prev_winner = winner_flat
        tm.prev_distal = distal
        tm.step_index += 1
        return SimpleNamespace(active_column=active, winner_flat=winner_flat,
                               cell_activation=activation, cell_prediction=prediction,     # prediction: own columns only
                               bursting=bursting, distal_state=distal)

    def _scan_owned(self, activation, step):
        """PredictiveProjection.process restricted to the segments this rank owns."""
        tm, p = self.tm, self.tm.params
        act_pad = tm._padded(activation)
        segs = np.arange(tm.S)
        owned = self.owns_cell(tm.seg_cell[:tm.S])
        _, _, idx = tm._targets(segs)
        hit = act_pad[idx] & owned[:, None]
        potential = hit.sum(axis=1).astype(np.int64)
        matching = np.flatnonzero(potential >= p.segment_matching_threshold) if p.segment_matching_threshold > 0 \
            else np.flatnonzero(owned)
        connected = (tm.perm[matching] >= tm.d.threshold32) & hit[matching]
        activation_count = connected.sum(axis=1).astype(np.int64)
        active = activation_count >= p.segment_activation_threshold
        cells = tm.seg_cell[matching]
        prediction = np.bincount(cells, weights=active, minlength=tm.N).astype(np.float64)
        u = draw_unit(tm.seed, STREAM_SEGMENT_JITTER, step, matching)
        jittered = (potential[matching].astype(np.float32).astype(np.float64) + u).astype(np.float32)
        cell_max = np.zeros(tm.N, dtype=np.float32)
        np.maximum.at(cell_max, cells, jittered)
        return SimpleNamespace(prediction=prediction, segment_potential=potential, matching_segment=matching,
                               matching_segment_activation=activation_count, matching_segment_active=active,
                               max_jittered_potential=cell_max, matching_segment_jittered_potential=jittered)

    def step(self, input_bits, all_gather, learning=True):
        """all_gather(record) -> [record of rank 0, ..., record of rank R-1]."""
        input_bits = np.asarray(input_bits, dtype=np.bool_)
        rec = self.local_record(input_bits)
        return self.finish_step(input_bits, all_gather(rec), learning=learning), rec


# ---- fixed-size wire format of one rank's record (what the HIP engine puts in its send buffer)

DEAD_CAP = 256


HOT_NONE = 0xFFFFFFFF


def _hot_offset(cap):
    return (cap * 20 + 4 + 4 * DEAD_CAP + 16 + 7) // 8 * 8


def record_nbytes(cap):
    """[boosted f64 x CAP][column | bursting << 31  u32 x CAP][winner word u32 x CAP][needs-a-segment word u32 x CAP]
    [n_dead u32][dead ids u32 x DEAD_CAP][n u32: candidate slots in use][n_hot u32][hot floor, 8 bytes: every own candidate at
    or above it is in the hot list (here the float64; the HIP engine sends its select key)][pad to 8 bytes]
    [hot boosted f64 x CAP][hot slot u16 x CAP][pad to 16 bytes]; the boosted overlaps of the CAP - n free slots and of the
    CAP - n_hot free hot entries have all bits set; n_hot = HOT_NONE: no hot list"""
    return (_hot_offset(cap) + cap * 10 + 15) // 16 * 16


def pack_record(rec, cell_dim, cap=None):
    n = len(rec.boosted)
    cap = n if cap is None else cap
    if n > cap:
        raise OverflowError("more candidates than the exchange record has slots")
    buf = np.zeros(record_nbytes(cap), dtype=np.uint8)
    weights = (np.uint32(1) << np.arange(cell_dim, dtype=np.uint32))
    o = 0
    buf[o:o + 8 * n] = rec.boosted.astype(np.float64).view(np.uint8)
    buf[o + 8 * n:o + 8 * cap] = 0xFF               # free slots: all bits set (CAND_PAD), what the HIP global select keys on
    o += 8 * cap
    colword = rec.col.astype(np.uint32) | (rec.bursting.astype(np.uint32) << np.uint32(31))
    buf[o:o + 4 * n] = colword.view(np.uint8); o += 4 * cap
    for mat in (rec.win, rec.unacc):
        buf[o:o + 4 * n] = (mat.astype(np.uint32) * weights).sum(axis=1).astype(np.uint32).view(np.uint8)
        o += 4 * cap
    if len(rec.dead) > DEAD_CAP:
        raise OverflowError("more newly dead segments than the exchange record holds")
    buf[o:o + 4] = np.array([len(rec.dead)], dtype=np.uint32).view(np.uint8); o += 4
    buf[o:o + 4 * len(rec.dead)] = rec.dead.astype(np.uint32).view(np.uint8); o += 4 * DEAD_CAP
    hot_slot = getattr(rec, "hot_slot", None)
    n_hot = HOT_NONE if hot_slot is None or cap > 65536 else len(hot_slot)
    buf[o:o + 8] = np.array([n, n_hot], dtype=np.uint32).view(np.uint8)
    buf[o + 8:o + 16] = np.array([getattr(rec, "hot_floor", 0.0)], dtype=np.float64).view(np.uint8)
    if n_hot != HOT_NONE:
        o = _hot_offset(cap)
        buf[o:o + 8 * n_hot] = rec.boosted[hot_slot].astype(np.float64).view(np.uint8)
        buf[o + 8 * n_hot:o + 8 * cap] = 0xFF
        o += 8 * cap
        buf[o:o + 2 * n_hot] = np.asarray(hot_slot).astype(np.uint16).view(np.uint8)
    return buf


def unpack_record(buf, cap, cell_dim):
    n, n_hot = (int(v) for v in buf[cap * 20 + 4 + 4 * DEAD_CAP:][:8].view(np.uint32))
    hot_floor = float(buf[cap * 20 + 4 + 4 * DEAD_CAP + 8:][:8].view(np.float64)[0])
    assert np.all(buf[8 * n:8 * cap] == 0xFF), "free candidate slots must carry the pad value"
    o = 0
    boosted = buf[o:o + 8 * n].view(np.float64).copy(); o += 8 * cap
    colword = buf[o:o + 4 * n].view(np.uint32); o += 4 * cap
    mats = []
    for _ in range(2):
        words = buf[o:o + 4 * n].view(np.uint32); o += 4 * cap
        mats.append(((words[:, None] >> np.arange(cell_dim, dtype=np.uint32)) & 1).astype(np.bool_))
    n_dead = int(buf[o:o + 4].view(np.uint32)[0]); o += 4
    dead = buf[o:o + 4 * n_dead].view(np.uint32).astype(np.int64)
    hot_slot = None
    if n_hot != HOT_NONE:
        o = _hot_offset(cap)
        assert np.all(buf[o + 8 * n_hot:o + 8 * cap] == 0xFF), "free hot entries must carry the pad value"
        hot_slot = buf[o + 8 * cap:o + 8 * cap + 2 * n_hot].view(np.uint16).astype(np.int64)
        assert np.array_equal(buf[o:o + 8 * n_hot].view(np.float64).view(np.int64), boosted[hot_slot].view(np.int64))
    return SimpleNamespace(boosted=boosted, col=(colword & np.uint32(0x7FFFFFFF)).astype(np.int64),
                           bursting=(colword >> np.uint32(31)).astype(np.bool_), win=mats[0], unacc=mats[1], dead=dead,
                           hot_slot=hot_slot, hot_floor=hot_floor)
