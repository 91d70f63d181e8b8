// Temporal Memory roles and kernels: activation, the middle of the step (lists, allocation, classification),
// learning and growth, the segment scan.
// Part of the single translation unit htm_engine.hip (included there, in this order:
// htm_dev.h, htm_sp_kernels.h, htm_tm_kernels.h, htm_pipeline.h).
#ifndef BITHTM_HTM_TM_KERNELS_H
#define BITHTM_HTM_TM_KERNELS_H

// ------------------------------------------------------------------------------------------
// Temporal Memory

// stand-alone TM: take the active columns from the caller, clear the per-step words
__global__ __launch_bounds__(256) void k_tm_load_active(Dev d, int p, const int *cols, int n) {
    for (int c = blockIdx.x * 256 + threadIdx.x; c < d.C * d.WPC; c += gridDim.x * 256) {
        d.act[p][c] = 0;
        d.pred[p][c] = 0;
        d.win[p][c] = 0;
        if (c < n) d.active_cols[p][c] = cols[c];
        if (c < d.colwords) d.colbits[p][c] = 0;
    }
}

// stand-alone TM: per-column activation, one active column per lane group
__global__ __launch_bounds__(256) void k_tm_activate(Dev d, int p, int n_active, int want_winner) {
    const int idx = blockIdx.x * tm_groups_per_block(d) + tm_group_of(d, threadIdx.x);
    const bool ok = idx < n_active;
    const int a = ok ? d.active_cols[p][idx] : 0;
    if (ok && (threadIdx.x & (d.KP - 1)) == 0) atomicOr(&d.colbits[p][a >> 5], 1u << (a & 31));
    tm_activate_column(d, p, want_winner, ok, a, idx, tm_pred_words(d, p, ok, a));
}

// PredictiveProjection.update called on its own (htm_tm_update): the learning cells come from the caller, grouped by
// column -- pass 0 clears the step's winner words, pass 1 stores the n columns' lists and words where the middle launch
// looks for them
// (winw / unacc: WPC words per listed column)
__global__ __launch_bounds__(256) void k_tm_ext_winners(Dev d, int p, const int *cols, const uint32_t *winw, const uint32_t *unacc, int n, int pass) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (pass == 0) { for (int c = i; c < d.C * d.WPC; c += gridDim.x * 256) d.win[p][c] = 0; return; }
    if (i < n * d.WPC) {
        const int col = cols[i / d.WPC], wi = col * d.WPC + i % d.WPC;
        if (i % d.WPC == 0) d.active_cols[p][i / d.WPC] = col;
        d.actw_id[i] = wi;
        d.win[p][wi] = winw[i];
        d.winw_idx[i] = winw[i];
        d.unacc_word[i] = unacc[i];
        d.actcnt[i] = 0;
    }
}

// PredictiveProjection.process called on its own (htm_tm_scan): the active cells come from the caller as one word per
// column; the column bitmap, the count of active cells and clean accumulators for the scan
// (actw: WPC words per column)
__global__ __launch_bounds__(256) void k_tm_ext_active(Dev d, int p, const uint32_t *actw) {
    for (int c0 = (blockIdx.x * 256 + (threadIdx.x & ~63)); c0 < ((d.C + 63) & ~63); c0 += gridDim.x * 256) {
        const int c = c0 + lane_id();
        uint32_t any = 0, cnt = 0;
        for (int h = 0; h < d.WPC; ++h) {
            const uint32_t w = c < d.C ? actw[c * d.WPC + h] & cell_mask_word(d.K, h) : 0u;
            if (c < d.C) { d.act[p][c * d.WPC + h] = w; d.pred[p][c * d.WPC + h] = 0; }
            any |= w;
            cnt += (uint32_t)__popc(w);
        }
        const u64 m = __ballot(any != 0);
        if (lane_id() == 0) *(u64 *)&d.colbits[p][c0 >> 5] = m;
        const uint32_t cells = wave_sum(cnt);
        if (lane_id() == 0 && cells) atomicAdd(&d.ctr->n_active_cells, (int)cells);
    }
}

__device__ __forceinline__ bool col_is_local(const Dev &d, int cell) { const int col = cell >> d.LK; return col >= d.c0 && col < d.c1; }

// the global id of a local row (unsharded handles: the row IS the id)
__device__ __forceinline__ int seg_gid_of(const Dev &d, int row) { return d.seg_gid ? d.seg_gid[row] : row; }

// bind segment `seg` to winner cell `cell` (projections.py:275-281) and queue it for learning at
// work-list slot `pos`; rows are packed, so clearing a recycled row (projections.py:82-85) is nsyn = 0.
// Unsharded handles (row == id).
__device__ __forceinline__ void tm_bind_segment(const Dev &d, int seg, int cell, bool recycled, int pos) {
    // (the recyclable counts per 1024 ids are settled by the caller for all the ids it binds, once per 1024-block:
    // what becomes of an empty row is known before the learning role has grown it, and same-address atomics are slow)
    if (recycled) atomicSub(&d.segcount[d.seg_cell[seg]], 1);
    d.seg_cell[seg] = cell;
    d.seg_nsyn[seg] = (int)SEG_BUSY;
    atomicAdd(&d.segcount[cell], 1);
    if (pos < d.work_cap) d.work[pos] = (uint32_t)seg; else atomicOr(&d.ctr->error, 4);
}

// Column-sharded handles: request `rank` of this step gets global id `gid` (recycled: an id whose dead bit is set).
// Every rank takes the identical decision and keeps the replicated dead bits / counts in step; the rank that owns
// the id's old row gives it up, the rank that owns `cell` takes a local row for it and queues that row for the
// learning role.  `grown`: the synapses the row will have once that role has run (projections.py:114-127 on an
// empty row: min(sampling, previous winners)) -- known everywhere, so the bits can be set now.
// pass 0: give up rows (pushes on the free stack); pass 1: take rows (pops) -- the two are separated by a barrier.
__device__ __forceinline__ void shard_bind(const Dev &d, int p, int gid, int cell, bool recycled, int grown, int pass) {
    Counters *c = d.ctr;
    const bool to_me = col_is_local(d, cell);
    const int old_row = recycled ? d.g2l[gid] : -1;                 // >= 0: the id was mine
    if (pass == 0) {
        const bool alive = grown >= d.match_thr;
        if (recycled && alive) { atomicAnd(&d.dead_bits[gid >> 5], ~(1u << (gid & 31))); recyc_add(d, gid >> 10, -1); }
        if (!recycled && !alive) { atomicOr(&d.dead_bits[gid >> 5], 1u << (gid & 31)); recyc_add(d, gid >> 10, 1); }
        if (old_row >= 0) {
            atomicSub(&d.segcount[d.seg_cell[old_row]], 1);
            if (!to_me) {                                           // the id moves to another rank: free the row
                d.seg_gid[old_row] = -1;
                d.seg_nsyn[old_row] = 0;
                d.g2l[gid] = -1;
                d.lfree[atomicAdd(&c->n_lfree, 1)] = old_row;
            }
        }
        return;
    }
    if (!to_me) return;
    int row = old_row;
    if (row < 0) {
        const int top = atomicSub(&c->n_lfree, 1) - 1;              // a freed row if there is one, else a fresh one
        if (top >= 0) {
            row = d.lfree[top];
        } else {
            atomicAdd(&c->n_lfree, 1);
            row = atomicAdd(&c->L, 1);
        }
        if (row >= d.Lcap) { atomicOr(&c->error, 1); return; }
        d.seg_gid[row] = gid;
        d.g2l[gid] = row;
    }
    d.seg_cell[row] = cell;
    d.seg_nsyn[row] = (int)SEG_BUSY;
    atomicAdd(&d.segcount[cell], 1);
    const int pos = atomicAdd(&c->n_work[p], 1);
    if (pos < d.work_cap) d.work[pos] = (uint32_t)row; else atomicOr(&c->error, 4);
}

// DenseProjection.update (projections.py:23-24) on winner row ri, fused with the rebuild of that
// row's connected mask, by TPR threads (t = 0..TPR-1): two float64 per lane (16-byte accesses); a
// wave covers 128 consecutive elements = four mask words, assembled from the ballots of its even
// and odd elements
// diagnostic build (-DBITHTM_ROWS_STAMPS, handle created under BITHTM_TRACE=1): device clock at the phases of the first 1 024
// row blocks, d.trace[7 * 8192 + row index * 8 + phase] (tools/rows_phases.py)
#ifdef BITHTM_ROWS_STAMPS
#define ROW_STAMP(i) do { if (d.trace && t == 0 && ri < 1024) d.trace[(size_t)7 * 8192 + (size_t)ri * 8 + (i)] = wall_clock64(); } while (0)
#else
#define ROW_STAMP(i) do { } while (0)
#endif
// p: parity of the step the rows belong to; ahead = 1 when that step's index is not published yet
// (the row update runs beside the previous step's scan): it is step[p ^ 1] + 1 then
// OWN (two-launch schedule, k_act_mid_rows; unsharded, the windowed select): the block also does, for its column, what the
// overlap role of the same launch does for the columns that did not win (role_overlap, fold = 2) -- step p's duty-cycle update
// (regularizations.py:19-21: this column won), the overlap of the NEW connected bits with the coming step's input
// (projections.py:18-21), boosted overlap and key (regularizations.py:15-17), the key's bin of the select histogram.  The new
// mask words are in the ballots of the pass: no other block has to wait for them.
template <int TPR, bool OWN = false>
__device__ __forceinline__ void role_sp_row(const Dev &d, int p, const uint32_t *__restrict__ bank, int n_inputs, int ahead, int ri, int t) {
    ROW_STAMP(0);
    const uint32_t step = ahead ? d.ctr->step[p ^ 1] + 1u : d.ctr->step[p];
    const uint32_t *in = bank + (size_t)(step % (uint32_t)n_inputs) * d.W;
    const uint32_t *in_next = bank + (size_t)((step + 1u) % (uint32_t)n_inputs) * d.W;
    const int row = d.active_cols[p][ri];
    if (row < d.c0 || row >= d.c1) return;          // another rank's column
    ROW_STAMP(1);
    double *prow = d.perm + (size_t)row * d.Ipad;
    uint32_t *mrow = d.mask + (size_t)row * d.W;
    float duty0 = 0.f;                              // (asked for now, used when the row is done)
    uint32_t wbase = 0u;
    if (OWN && t == 0) {
        duty0 = d.duty[row];
        wbase = d.ctr->sel_win[p ^ 1];
    }
    int own_cn = 0;
    for (int i0 = 0; i0 < d.Ipad; i0 += 2 * TPR) {
        const int e0 = i0 + 2 * t;                   // Ipad is a multiple of 128: e0 + 1 < Ipad whenever e0 < Ipad
        bool c0 = false, c1 = false;
        if (e0 < d.Ipad) {
            // (one pass at a time on purpose: with the second pass's loads in flight beside the first's the first pass waits
            // 2.9 us instead of 1.8 and the launch is a microsecond longer -- 1 311 scattered 8-KB rows read at 3-3.7 TB/s
            // whatever is asked at once: tools/rows_phases.py; again in the two-launch schedule's first launch: 43.65 against 43.85 k
            // timesteps/s, twice)
            double2 v = *(double2 *)(prow + e0);
            const uint32_t bits = in[e0 >> 5] >> (e0 & 31);
            // (OWN: the coming input's bits of the thread's two elements, asked for with the row -- the lane that stores the mask
            // words fetching the coming input's words once it had them cost every pass a round trip: 1.6 us instead of 0.76)
            const uint32_t nbits = OWN ? in_next[e0 >> 5] >> (e0 & 31) : 0u;
            if (e0 < d.I) { v.x = v.x + ((bits & 1u) ? d.sp_don : d.sp_doff); c0 = v.x >= d.sp_thr; }
            if (e0 + 1 < d.I) { v.y = v.y + ((bits & 2u) ? d.sp_don : d.sp_doff); c1 = v.y >= d.sp_thr; }
            if (OWN) own_cn += (int)(c0 && (nbits & 1u)) + (int)(c1 && (nbits & 2u));
            if (i0 == 0) ROW_STAMP(2); else ROW_STAMP(4);       // (the pass's values are here)
            {   // 16-byte write-through store (sc0 sc1): nothing of the rows stays dirty in L2 for the kernel-end release
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                union { double2 d2; u32x4 u4; } cvt;
                cvt.d2 = v;
                double *dst = prow + e0;
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(cvt.u4) : "memory");
            }
        }
        const u64 b0 = __ballot(c0), b1 = __ballot(c1);
        const int base = i0 + 2 * (t & ~63);
        if (lane_id() == 0 && base < d.Ipad) {
            u64 *mw = (u64 *)&mrow[base >> 5];
            const u64 m0 = spread32((uint32_t)b0) | (spread32((uint32_t)b1) << 1);
            const u64 m1 = spread32((uint32_t)(b0 >> 32)) | (spread32((uint32_t)(b1 >> 32)) << 1);
            mw[0] = m0;
            mw[1] = m1;
        }
        if (i0 == 0) ROW_STAMP(3); else ROW_STAMP(5);
    }
    if (OWN) {
        __shared__ int s_own[TPR / 64];
        own_cn = (int)wave_sum((uint32_t)own_cn);
        if (lane_id() == 0) s_own[t >> 6] = own_cn;
        __syncthreads();
        if (t == 0) {
            int cn = 0;
#pragma unroll
            for (int wv = 0; wv < TPR / 64; ++wv) cn += s_own[wv];
            const int sp = p ^ 1;
            float dc = duty0 * d.mom;               // float32, two separately rounded operations
            dc = dc + d.dinc;
            d.duty[row] = dc;
            d.overlap[sp][row] = cn;
            const float f = htm_exp_f32(d.coef * dc);
            const double bo = (double)f * (double)cn;
            d.boosted[sp][row] = bo;
            const u64 key = select_key(bo);
            d.key[sp][row] = key;
            const uint32_t bin = win_bin(key, wbase);
            if (bin != 0u) {                        // (as role_overlap's flush, straight to the copies)
                atomicAdd(d.hist0 + (size_t)sp * HIST0_PAR + (size_t)(ri & (HIST_REP - 1)) * SEL_BINS + bin, 1u);
                atomicAdd(d.hist0 + (size_t)sp * HIST0_PAR + HIST0_FINE + (size_t)(ri & (COARSE_REP - 1)) * COARSE_STRIDE + (bin >> 6), 1u);
            }
        }
    }
}

// The middle of TemporalMemory.process / PredictiveProjection.update, one launch:
//   block 0      ordered lists of winner cells (networks.py:103-104) and of winners that need a new
//                segment (projections.py:271-273); SparseProjection.add_output (projections.py:79-95):
//                recycle the lowest-id segments with fewer than matching_threshold synapses, append
//                the rest; bind them to the winners in ascending cell order (:275-281)
//   blocks 1..   which previous matching segments learn, which are punished (projections.py:264-269;
//                punishment mask built at networks.py:107-108,111)
//   last n_sp_rows blocks   DenseProjection.update (projections.py:23-24) on one winner row each,
//                fused with the rebuild of that row's connected mask: independent of the TM work and
//                bandwidth-bound, it rides along with the latency-bound block 0
// block 0 and the classify blocks 1..n_cls of the middle launch (below)
// same_launch: the activation ran in THIS launch (k_act_mid_rows, behind its fan-in) and so do the clears of the dense words of
// inactive columns -- a winner bit counts only on a column of the step's bitmap (what the cleared words say everywhere else),
// and what the activation wrote is read with agent-scope loads: past this XCD's L2, which may hold the lines as they were.
// (An acquire fence instead -- buffer_inv sc1 by each of the role's 1 540 waves -- took the launch from 11 to 26 us.)
template <typename T>
__device__ __forceinline__ T ld_agent(const T *ptr, int same_launch) {
    return same_launch ? __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *ptr;
}
// wait: called once by the whole block, right before the first read of anything the activation wrote -- k_act_mid_rows passes its
// fan-in there; everything the role can fetch and decide WITHOUT the activation's output is done before (a classification
// block: the match words, its first rows' owner cells, jitter, prediction bits and per-cell maxima -- three round trips that
// now run under the activation instead of behind it)
struct NoWait { __device__ __forceinline__ void operator()() const {} };
template <int BS, typename Wait = NoWait>
__device__ __forceinline__ void role_mid(const Dev &d, int p, int n_active, int want_winner, int learning, int blk, int n_cls, int same_launch = 0,
                                         Wait wait = Wait()) {
    Counters *c = d.ctr;
    // (the classification blocks' list of non-zero match words; block 0 keeps the 1 024-blocks it recycles from in s_words)
    __shared__ int s_words[BS];
    __shared__ uint32_t s_bits[BS];
    __shared__ int s_nwords;
    if (blk > 0) {
        // Which of the previous matching segments learn, which are punished.  The scan left one bit per row; the
        // items of a wave are appended with one reservation (same-address atomics are slow: ~88 per us).
        if (!learning || !c->has_distal) return;
        const int q = p ^ 1;
        const int n = d.world > 1 ? c->L : c->S;     // rows at or above the count of the last scan are not matching
        // classification of one matching row, in two halves: what does not need the step's winner cells (cand: the row
        // learns if its cell is a winner), then the winner bit
        bool waited = false;
        auto wait_once = [&]() { if (!waited) { wait(); waited = true; } };           // (block-uniform call sites)
        auto gather = [&](int seg, bool &cand, bool &punish, int &cell) {
            const uint32_t info = d.seg_info[seg];
            cell = d.seg_cell[seg];
            const float jit = d.seg_jit[seg];
            const int cw = cell >> 5, cb = cell & 31, col = cell >> d.LK;               // (the cell's word in the dense arrays; its column)
            const bool on_col = (d.colbits[p][col >> 5] >> (col & 31)) & 1u;
            const bool unpred = !((d.pred[q][cw] >> cb) & 1u);                           // :266
            const bool best = fabsf(jit - __uint_as_float(d.cellmax[q][cell])) < d.eps;  // :267
            cand = ((info >> 31) || (unpred && best)) && (on_col || !same_launch);       // :268, but for the winner bit
            // (a column is inactive: not on the step's column bitmap -- a column of 64 cell slots may be active with one of
            // its two active words empty)
            punish = d.punish ? (d.punish[cw] >> cb) & 1u : !on_col;      // :269
        };
        auto decide = [&](bool cand, int cell) {                                          // :268
            return cand && ((ld_agent(&d.win[p][cell >> 5], same_launch) >> (cell & 31)) & 1u);
        };
        // (... up to 8 rows per thread; 2 where the role's blocks wait for the activation of their own launch: few blocks, whatever they
        // do after the wait is at the end of the launch's longest chain)
        if (n <= (d.cls_rows_max >= 0 ? d.cls_rows_max : (same_launch ? 2 : 8) * n_cls * BS)) {
            // small pools: one row per thread, so that the rows of a word -- segments created together match together --
            // are classified side by side, not one after the other
            for (int i0 = (blk - 1) * BS; i0 < n; i0 += n_cls * BS) {
                const int seg = i0 + (int)threadIdx.x;
                bool cand = false, punish = false;
                int cell = 0;
                if (seg < n && ((d.match_bits[q][seg >> 5] >> (seg & 31)) & 1u)) gather(seg, cand, punish, cell);
                wait_once();
                const bool learn = decide(cand, cell);
                if (learn || punish) d.seg_nsyn[seg] |= (int)SEG_BUSY;
                const u64 ml = __ballot(learn), mp = __ballot(punish);
                const int n_l = __popcll(ml), n_p = __popcll(mp);
                if (n_l + n_p == 0) continue;
                int base = 0;
                if (lane_id() == 0) base = atomicAdd(&c->n_work[p], n_l + n_p);
                base = wave_read(base, 0);
                if (learn) {
                    const int pos = base + __popcll(ml & lanemask_lt());
                    if (pos < d.work_cap) d.work[pos] = (uint32_t)seg; else atomicOr(&c->error, 4);
                }
                if (punish) {
                    const int pos = base + n_l + __popcll(mp & lanemask_lt());
                    if (pos < d.work_cap) d.work[pos] = (uint32_t)seg | 0x80000000u; else atomicOr(&c->error, 4);
                }
            }
            return;
        }
        // Large pools: the match bits are read a 32-row word per thread (almost all of them zero) -- but the rows of a word are
        // NOT walked by the thread that read it: segments created together match together, a learned pattern's ~1 300 of them
        // are 40 consecutive words with nearly every bit set, and a thread walking one chained 64 dependent round trips
        // (38 us for this launch on a learned pool of 1.5 M segments).  Consecutive words go to different BLOCKS, a block
        // lists its few non-zero words in LDS, and the listed words are classified 32 rows by 32 lanes, eight words per
        // pass, exactly as the small-pool form does.  Which block: the 32 words of a 128-byte line go to 32 different
        // blocks of ONE residue mod 8 -- blocks are dealt round-robin to the 8 XCDs, so a line is fetched into one L2, not
        // eight (a pool of 134 M rows has 17 MB of match bits).  With u = the index of a word among those of the lines
        // L = x mod 8: block 8 * (u mod G8) + x takes it as its (u / G8)-th word.
        const int nwords = (n + 31) >> 5;
        const bool by_xcd = n_cls % 8 == 0 && n_cls >= 256;
        const int G8 = by_xcd ? n_cls / 8 : 1, x = (blk - 1) & 7, g = (blk - 1) >> 3;
        const int per_class = ((nwords + 255) >> 8) * 32;          // words of one residue class (rounded up to whole lines)
        const int n_turns = by_xcd ? (per_class + G8 - 1) / G8 : (nwords + n_cls - 1) / n_cls;
        for (int t0 = 0; t0 < n_turns; t0 += BS) {
            if (threadIdx.x == 0) s_nwords = 0;
            __syncthreads();
            const int T = t0 + (int)threadIdx.x;
            int w;
            if (by_xcd) {
                const int u = T * G8 + g;
                w = ((x + 8 * (u >> 5)) << 5) + (u & 31);
            } else {
                w = (blk - 1) + n_cls * T;
            }
            const uint32_t word = (T < n_turns && w < nwords) ? d.match_bits[q][w] : 0u;
            if (word) {
                const int slot = atomicAdd(&s_nwords, 1);
                s_words[slot] = w;
                s_bits[slot] = word;
            }
            __syncthreads();
            const int listed = s_nwords;
            for (int i0 = 0; i0 < listed; i0 += BS / 32) {
                const int i = i0 + ((int)threadIdx.x >> 5);
                const bool on = i < listed && ((s_bits[min(i, BS - 1)] >> (threadIdx.x & 31)) & 1u);
                const int seg = on ? s_words[i] * 32 + (int)(threadIdx.x & 31) : 0;
                bool cand = false, punish = false;
                int cell = 0;
                if (on) gather(seg, cand, punish, cell);
                wait_once();
                const bool learn = decide(cand, cell);
                if (learn || punish) d.seg_nsyn[seg] |= (int)SEG_BUSY;
                const u64 ml = __ballot(learn), mp = __ballot(punish);
                const int n_l = __popcll(ml), n_p = __popcll(mp);
                if (n_l + n_p == 0) continue;
                int base = 0;
                if (lane_id() == 0) base = atomicAdd(&c->n_work[p], n_l + n_p);
                base = wave_read(base, 0);
                if (learn) {
                    const int pos = base + __popcll(ml & lanemask_lt());
                    if (pos < d.work_cap) d.work[pos] = (uint32_t)seg; else atomicOr(&c->error, 4);
                }
                if (punish) {
                    const int pos = base + n_l + __popcll(mp & lanemask_lt());
                    if (pos < d.work_cap) d.work[pos] = (uint32_t)seg | 0x80000000u; else atomicOr(&c->error, 4);
                }
            }
        }
        return;
    }
    // ---- block 0
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_cells;
    __shared__ int s_nneed;
    if (threadIdx.x == 0) { s_cells = 0; s_nneed = 0; }
    const int S = c->S, nb = (S + 1023) >> 10;
    // first batch of the per-1024-segment recyclable counts: fetched with the column lists, not after them
    const uint32_t recyc_first = (int)threadIdx.x < nb ? (uint32_t)d.recyc_cnt[threadIdx.x] : 0u;
    uint32_t carry_w = 0, carry_u = 0, n_cells = 0;
    // LPT consecutive columns per thread, one scan per pass: with 256 threads one pass covers 2048 winner
    // columns, so every load of the lists is in flight at once (under the load of the row updates that
    // share this launch a dependent round trip costs about 3 us)
    constexpr int LPT = 8;
    static_assert(LPT == 8, "the list pass loads 8 entries per thread");
    // (the lists are per WORD of the active columns: one per column, two where a column has 64 cell slots; a[] = the word's index
    // in the dense arrays, so that word * 32 + bit is the cell)
    const int n_slots = n_active * d.WPC;
    // what the allocation below needs of the PREVIOUS step and of the pool, asked for before the wait: the counts of recyclable
    // segments per 2^20 ids (their first 256), whether and how many winners the last step had
    const int nb2 = (nb + 1023) >> 10;
    const int cnt2_first = ((int)threadIdx.x < 256 && (int)threadIdx.x < nb2) ? d.recyc_cnt2[threadIdx.x] : 0;
    const int prev_has_winner = c->has_winner[p ^ 1], prev_n_win = c->n_win[p ^ 1];
    wait();
    for (int base = 0; base < n_slots; base += LPT * BS) {
        const int i0 = base + LPT * (int)threadIdx.x;
        uint32_t ww[LPT], uw[LPT];                 // (counts are recomputed from the words: the launch is capped at 64 registers)
        int a[LPT];
        uint32_t vsum = 0;
        u64 ac = 0;
        if (i0 < n_slots && same_launch) {         // (agent-scope loads, 8 bytes each: the arrays are padded by 8 entries)
#pragma unroll
            for (int j = 0; j < LPT; j += 2) {
                const u64 av = __hip_atomic_load((const u64 *)(d.actw_id + i0 + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const u64 wv = __hip_atomic_load((const u64 *)(d.winw_idx + i0 + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const u64 uv = __hip_atomic_load((const u64 *)(d.unacc_word + i0 + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a[j] = (int)(uint32_t)av; a[j + 1] = (int)(uint32_t)(av >> 32);
                ww[j] = (uint32_t)wv; ww[j + 1] = (uint32_t)(wv >> 32);
                uw[j] = (uint32_t)uv; uw[j + 1] = (uint32_t)(uv >> 32);
            }
            ac = __hip_atomic_load((const u64 *)(d.actcnt + i0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (i0 < n_slots) {                 // 16-byte loads (the arrays are padded by 8 entries), masked below
            const int4 a0 = *(const int4 *)(d.actw_id + i0), a1 = *(const int4 *)(d.actw_id + i0 + 4);
            const uint4 w0 = *(const uint4 *)(d.winw_idx + i0), w1 = *(const uint4 *)(d.winw_idx + i0 + 4);
            const uint4 u0 = *(const uint4 *)(d.unacc_word + i0), u1 = *(const uint4 *)(d.unacc_word + i0 + 4);
            ac = *(const u64 *)(d.actcnt + i0);
            a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
            ww[0] = w0.x; ww[1] = w0.y; ww[2] = w0.z; ww[3] = w0.w; ww[4] = w1.x; ww[5] = w1.y; ww[6] = w1.z; ww[7] = w1.w;
            uw[0] = u0.x; uw[1] = u0.y; uw[2] = u0.z; uw[3] = u0.w; uw[4] = u1.x; uw[5] = u1.y; uw[6] = u1.z; uw[7] = u1.w;
        }
        if (i0 < n_slots) {
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                const bool ok = i0 + j < n_slots;
                if (!ok) { ww[j] = 0; uw[j] = 0; }
                n_cells += ok ? (uint32_t)((ac >> (8 * j)) & 0xFFu) : 0u;
            }
        } else {
#pragma unroll
            for (int j = 0; j < LPT; ++j) { a[j] = 0; ww[j] = 0; uw[j] = 0; }
        }
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            vsum += want_winner ? (uint32_t)__popc(ww[j]) | ((uint32_t)__popc(uw[j]) << 16) : 0u;    // winners | needing a segment
        }
        uint32_t total;
        uint32_t run = block_excl_scan<BS>(vsum, s_wave, total);
        if (want_winner) {
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                int pw = carry_w + (run & 0xFFFFu), pu = carry_u + (run >> 16);
                uint32_t w1 = ww[j], u1 = uw[j];
                while (w1) { int b = __ffs(w1) - 1; w1 &= w1 - 1; d.winners[p][pw++] = a[j] * 32 + b; }
                while (u1) { int b = __ffs(u1) - 1; u1 &= u1 - 1; d.unacc_list[pu++] = a[j] * 32 + b; }
                run += (uint32_t)__popc(ww[j]) | ((uint32_t)__popc(uw[j]) << 16);
            }
        }
        carry_w += total & 0xFFFFu;
        carry_u += total >> 16;
    }
    n_cells = wave_sum(n_cells);
    if (lane_id() == 0) atomicAdd(&s_cells, n_cells);
    lds_barrier();                                  // (LDS only: the winner lists' stores need not be acknowledged for the count to be read)
    const int n_un = (learning && c->has_distal) ? (int)carry_u : 0;
    if (threadIdx.x == 0) {
        c->n_win[p] = want_winner ? (int)carry_w : 0;
        c->has_winner[p] = want_winner;
        c->n_un = n_un;
        c->n_active_cells = (int)s_cells;
        if (n_un == 0) c->n_bind[p] = 0;
    }
    if (n_un == 0) return;
    uint32_t carry = 0;                       // recyclable segments seen so far
    // the counts per 2^20 ids first (one load): ranges without a recyclable segment are passed over -- a pool of 134 M
    // segments has 130 k per-1024 counts, and walking them 256 at a time took this block 0.3 ms per step
    // (the barriers of this part order LDS traffic only -- lds_barrier: none of them waits for the acknowledgement of the stores
    // before it; under the row traffic of the launch each such wait was a microsecond of this block's chain)
    __shared__ int s_cnt2[256];
    constexpr int NEED_LDS = BS / 2;               // needed 1 024-blocks kept in LDS (s_words, as pairs); more go through d.recyc_need
    for (int r0 = 0; r0 < nb2 && carry < (uint32_t)n_un; r0 += 256) {
        lds_barrier();
        s_cnt2[threadIdx.x & 255] = r0 == 0 ? cnt2_first : (((int)threadIdx.x < 256 && r0 + (int)threadIdx.x < nb2) ? d.recyc_cnt2[r0 + threadIdx.x] : 0);
        lds_barrier();
        for (int r = r0; r < min(r0 + 256, nb2) && carry < (uint32_t)n_un; ++r) {
            if (s_cnt2[r - r0] == 0) continue;
            for (int base = r << 10; base < min(nb, (r + 1) << 10); base += BS) {
                const int b = base + threadIdx.x;
                const uint32_t v = base == 0 ? recyc_first : ((b < nb && b < ((r + 1) << 10)) ? (uint32_t)d.recyc_cnt[b] : 0u);
                uint32_t total;
                const uint32_t ex = block_excl_scan<BS, true>(v, s_wave, total);
                if (v > 0 && carry + ex < (uint32_t)n_un) {
                    const int slot = atomicAdd(&s_nneed, 1);
                    if (slot < NEED_LDS) {
                        s_words[2 * slot] = b;
                        s_words[2 * slot + 1] = (int)(carry + ex);
                    } else {
                        d.recyc_need[2 * slot] = b;
                        d.recyc_need[2 * slot + 1] = (int)(carry + ex);
                    }
                }
                carry += total;
                if (carry >= (uint32_t)n_un) break;
            }
        }
    }
    lds_barrier();
    if (s_nneed > NEED_LDS) __syncthreads();       // (entries that went through memory: their stores acknowledged)
    const int n_r = min(n_un, (int)carry);
    int n_new = n_un - n_r;
    if (S + n_new > d.Scap) {
        if (threadIdx.x == 0) atomicOr(&c->error, 1);
        n_new = max(d.Scap - S, 0);
    }
    // the bound segments are queued from the back of the work array (no reservation to wait for);
    // sharded: only the binds to own cells are queued, each with its own reservation at the front
    const bool whole = d.world == 1;
    __syncthreads();                                // (this one for memory: the list of winners without a segment, stored by the list pass, is read below)
    const int wbase = d.work_cap - (n_r + n_new);
    const int n_w = prev_has_winner ? prev_n_win : -1;
    const int grown = n_w > 0 ? min(d.sample, n_w) : 0;
    const int n_need = s_nneed;
    for (int i = 0; i < n_need; ++i) {          // each needed 1024-block: rank its recyclable segments
        const int b = i < NEED_LDS ? s_words[2 * i] : d.recyc_need[2 * i], off = i < NEED_LDS ? s_words[2 * i + 1] : d.recyc_need[2 * i + 1];
        constexpr int IPT = 1024 / BS;             // consecutive segments per thread
        uint32_t fl[IPT], cnt = 0;
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const int seg = b * 1024 + (int)threadIdx.x * IPT + j;
            const bool dead = whole ? d.seg_nsyn[min(seg, d.Scap - 1)] < d.match_thr : (d.dead_bits[seg >> 5] >> (seg & 31)) & 1u;
            fl[j] = (seg < S && dead) ? 1u : 0u;
            cnt += fl[j];
        }
        uint32_t total;
        int rank = off + (int)block_excl_scan<BS>(cnt, s_wave, total);
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const int seg = b * 1024 + (int)threadIdx.x * IPT + j;
            if (fl[j] && rank < n_r) {
                if (whole) tm_bind_segment(d, seg, d.unacc_list[rank], true, wbase + rank);
                else d.asg_gid[rank] = seg;
            }
            rank += (int)fl[j];
        }
        // the ids recycled out of this 1024-block leave its recyclable count if the learning role is going to grow
        // them past the matching threshold (known now: an empty row grows `grown` synapses) -- one update per block
        if (whole && threadIdx.x == 0 && grown >= d.match_thr) recyc_add(d, b, -max(0, min((int)total, n_r - off)));
    }
    if (whole) {
        // (the cells of eight bindings per thread first, clamped and unconditional, then their stores: with one load per
        // iteration the compiler waits for everything outstanding, the previous iteration's stores and atomics included --
        // six round trips while every column bursts)
        for (int i0 = 0; i0 < n_new; i0 += 8 * BS) {
            int cl[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) cl[j] = d.unacc_list[n_r + min(i0 + j * BS + (int)threadIdx.x, n_new - 1)];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * BS + (int)threadIdx.x;
                if (i < n_new) tm_bind_segment(d, S + i, cl[j], false, wbase + n_r + i);
            }
        }
        if (n_new > 0 && grown < d.match_thr)      // fresh ids [S, S + n_new) that will stay below the matching threshold
            for (int b = (S >> 10) + (int)threadIdx.x; b <= (S + n_new - 1) >> 10; b += BS)
                recyc_add(d, b, min(S + n_new, (b + 1) << 10) - max(S, b << 10));
    } else {                                    // sharded: ids first, then rows given up, then rows taken
        __syncthreads();
        for (int pass = 0; pass < 2; ++pass) {
            for (int i = threadIdx.x; i < n_r + n_new; i += BS)
                shard_bind(d, p, i < n_r ? d.asg_gid[i] : S + (i - n_r), d.unacc_list[i], i < n_r, grown, pass);
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        c->n_recycled = n_r;
        c->n_new = n_new;
        c->n_bind[p] = whole ? n_r + n_new : 0;
        c->S_old = S;
        c->S = S + n_new;
    }
}

// SparseProjection.update_permanence (projections.py:97-109) and add_edge (:111-161) for one
// work item per wave.  Permanences: float64 sum, float32 store, prune on the float64 value; the
// surviving synapses are re-packed to the front of the row.  Growth: the n_add previous winner
// cells with the smallest keyed priority that the segment does not have yet.
template <int EPL, int BS>
struct LearnShared { u64 cand[BS / 64][CAND_CAP]; int keep[BS / 64][EPL * 64]; int win[WIN_LDS]; };
static_assert(CAND_CAP == 4 * 64, "the growth path holds the staged winners four per lane");

// diagnostic build (-DBITHTM_LEARN_STAMPS, handle created under BITHTM_TRACE=1): cycles per phase of the learning
// role, summed over all waves into the first words of the trace buffer (tools/learn_phases.py)
#ifdef BITHTM_LEARN_STAMPS
#define LSTAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); acc_[i] += now_ - stamp_; stamp_ = now_; } while (0)
#define LCOUNT(i, v) do { acc_[i] += (unsigned long long)(v); } while (0)
#else
#define LSTAMP(i) do { } while (0)
#define LCOUNT(i, v) do { } while (0)
#endif

// SELF: the wave also does the segment scan's work for its segment (potential and connected-active count against this
// step's active cells, from the synapses it holds in registers; publication as in role_scan) -- the schedule in which the
// learning role and the scan share a launch: the scan leaves the rows on the work list alone (SEG_BUSY).
template <int EPL, int BS, bool SELF = false>
__device__ __forceinline__ void role_learn(const Dev &d, int p, int blk, int nblk, LearnShared<EPL, BS> *sh) {
    int (*s_keep)[EPL * 64] = sh->keep;
    u64 (*s_cand)[CAND_CAP] = sh->cand;
    Counters *c = d.ctr;
    {   // forget the previous scan's per-cell maxima (sparse clear; every reader ran in an earlier
        // launch) and reset what the coming scan accumulates
        const int n = c->has_distal ? (d.world > 1 ? c->L : c->S) : 0;
        if (c->cm_dense_step == c->step[p] + 1u) {  // the step after a state import: everything
            for (int i = blk * BS + threadIdx.x; i < d.C * d.KP; i += nblk * BS) d.cellmax[p ^ 1][i] = 0u;
        } else if (n <= 8 * nblk * BS) {           // small pools: one row per thread (matching rows cluster in words)
            for (int i = blk * BS + threadIdx.x; i < n; i += nblk * BS)
                if ((d.match_bits[p ^ 1][i >> 5] >> (i & 31)) & 1u) d.cellmax[p ^ 1][d.seg_cell[i]] = 0u;
        } else {
            // large pools: one 32-row word per lane, almost all of them zero -- and a non-zero word's rows cleared by 32 lanes at
            // once, two words per pass (matching rows cluster: a lane walking the 32 set bits of its own word chained 32 round
            // trips -- each owner-cell load behind the previous store -- and the role's items started 15 us late).  Which
            // words a wave takes: as in role_mid's classification -- consecutive words to different waves, the words of a
            // 128-byte line to waves of blocks of one residue mod 8 (one XCD's L2 fetches the line).
            const int nwords = (n + 31) >> 5, lane = lane_id();
            constexpr int WPB = BS / 64;
            const bool by_xcd = nblk % 8 == 0 && (nblk / 8) * WPB >= 32;
            const int G8 = by_xcd ? (nblk / 8) * WPB : 1, x = blk & 7, g = (blk >> 3) * WPB + ((int)threadIdx.x >> 6);
            const int n_units = nblk * WPB, unit = blk * WPB + ((int)threadIdx.x >> 6);
            const int per_class = ((nwords + 255) >> 8) * 32;
            const int n_turns = by_xcd ? (per_class + G8 - 1) / G8 : (nwords + n_units - 1) / n_units;
            for (int t0 = 0; t0 < n_turns; t0 += 64) {
                const int T = t0 + lane;
                int w;
                if (by_xcd) {
                    const int u = T * G8 + g;
                    w = ((x + 8 * (u >> 5)) << 5) + (u & 31);
                } else {
                    w = unit + n_units * T;
                }
                const uint32_t word = (T < n_turns && w < nwords) ? d.match_bits[p ^ 1][w] : 0u;
                for (u64 nz = __ballot(word != 0); nz;) {
                    const int l0 = __ffsll((long long)nz) - 1;
                    nz &= nz - 1;
                    const int l1 = nz ? __ffsll((long long)nz) - 1 : l0;
                    const bool two = nz != 0;
                    nz &= nz - 1;
                    const uint32_t wa = wave_read(word, l0), wb = two ? wave_read(word, l1) : 0u;
                    const int ia = wave_read(w, l0), ib = wave_read(w, l1);
                    const uint32_t mine = lane < 32 ? wa : wb;
                    const int row = (lane < 32 ? ia : ib) * 32 + (lane & 31);
                    if ((mine >> (lane & 31)) & 1u) d.cellmax[p ^ 1][d.seg_cell[row]] = 0u;
                }
            }
        }
    }
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const int n_front = min(c->n_work[p], d.work_cap), n_back = c->n_bind[p];
    if (blk == 0 && threadIdx.x == 0) c->n_work_last = n_front + n_back;
    if (n_front + n_back > d.work_cap && blk == 0 && threadIdx.x == 0) atomicOr(&c->error, 4);
    const int n_work = min(n_front + n_back, d.work_cap);
    const uint32_t *act_prev = d.act[p ^ 1], *act_cur = d.act[p];
    const uint32_t base3 = htm_stream_base(d.seed, HTM_STREAM_SEGMENT_JITTER, c->step[p]);
    const int *winners = d.winners[p ^ 1];
    const int n_w = c->has_winner[p ^ 1] ? c->n_win[p ^ 1] : -1;       // -1: winner_input is None
    const uint32_t base2 = htm_stream_base(d.seed, HTM_STREAM_GROWTH, c->step[p]);
    // the previous winners, the population every growing segment samples from: once per block into LDS
    // (the staging pass of a growing segment otherwise waits for one global load per 64 winners)
    int *s_win = sh->win;
    for (int i = threadIdx.x; i < min(n_w, WIN_LDS); i += BS) s_win[i] = winners[i];
    __syncthreads();
    auto winner_at = [&](int i) -> int { return i < WIN_LDS ? s_win[i] : winners[i]; };
#ifdef BITHTM_LEARN_STAMPS
    unsigned long long acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // per wave, in registers; written once below
    unsigned long long stamp_ = __builtin_readcyclecounter();
#endif
    for (int item = blk * (BS / 64) + wv; item < n_work; item += nblk * (BS / 64)) {
        LSTAMP(0);
        const uint32_t w = d.work[item < n_front ? item : d.work_cap - n_back + (item - n_front)];
        const int seg = (int)(w & 0x7FFFFFFFu), mode = (int)(w >> 31);        // (a local row)
        const uint32_t gid = (uint32_t)seg_gid_of(d, seg);                    // the random draws are keyed by the global id
        const double dA = mode ? d.pun_act : d.lrn_act, dI = mode ? d.pun_inact : d.lrn_inact;
        const bool prune = mode ? d.pun_prune : d.lrn_prune;
        const int n = d.seg_nsyn[seg] & ~(int)SEG_BUSY;
        const int owner = SELF ? d.seg_cell[seg] : 0;                     // (with the count: behind the row's stores it would wait for them)
        int *prow = d.presyn + (size_t)seg * d.E;
        float *mrow = d.sperm + (size_t)seg * d.E;
        int n_keep = 0, n_active = 0;
        int pot = 0, conn = 0;                       // SELF: active presynaptic cells of THIS step (projections.py:247), connected ones (:171-172)
        // (the whole row first: a chunk's loads behind the previous chunk's stores would wait for those to complete)
        int ps_all[EPL];
        float pm_all[EPL];
#pragma unroll
        for (int jj = 0; jj < EPL; ++jj) {
            const int idx = jj * 64 + lane;
            // (unconditional: under `idx < n` the row's loads would wait for the synapse count, one more round trip on the
            // item's chain; slots past the count hold a free slot's -1 / -1.0 and are masked by `valid` below)
            ps_all[jj] = idx < d.E ? prow[idx] & SYN_CELL : 0;
            pm_all[jj] = idx < d.E ? mrow[idx] : 0.f;
        }
#pragma unroll
        for (int jj = 0; jj < EPL; ++jj) {
            const int idx = jj * 64 + lane;
            const bool valid = idx < n;
            const int ps = ps_all[jj];
            const float pm = pm_all[jj];
            const bool a = valid && ((act_prev[ps >> 5] >> (ps & 31)) & 1u);
            const bool a_now = SELF && valid && ((act_cur[ps >> 5] >> (ps & 31)) & 1u);
            const double p64 = (double)pm + (a ? dA : dI);               // :102-103
            const bool keep = valid && !(prune && p64 < 0.0);            // :105-108
            const float p32 = (float)p64;                                  // :104
            const bool connected = p32 >= d.perm_thr;
            const u64 mk = __ballot(keep);
            if (keep) {
                const int pos = n_keep + __popcll(mk & lanemask_lt());
                // (written through: the scan waves of this launch that take the row once its flag is gone may sit on another XCD)
                store_through(&prow[pos], ps | (connected ? (int)SYN_CONNECTED : 0));     // the scan's `permanence >= threshold`, kept with the id
                store_through(&mrow[pos], p32);
                s_keep[wv][pos] = ps;
            }
            n_keep += __popcll(mk);
            n_active += __popcll(__ballot(keep && a));                    // :114
            if (SELF) {
                pot += __popcll(__ballot(keep && a_now));
                conn += __popcll(__ballot(keep && a_now && connected));
            }
        }
        __builtin_amdgcn_wave_barrier();
        LSTAMP(1);
        LCOUNT(8, 1);
        int n_total = n_keep;
        if (mode == 0 && n_w > 0) {
            const int n_add = min(max(d.sample - n_active, 0), min(d.sample, n_w));     // :115
            if (n_add > 0) {
                // threshold T with n_add <= |{absent winners with priority < T}| <= CAND_CAP
                uint32_t lo = 0, hi = 1u << 24, T = 1u << 24;
                if (n_w > CAND_CAP) {
                    // expected number staged: 1.5 n_add + 12 (60 of them for a whole sample of 32: they fit one
                    // register slot per lane below, and fewer than n_add turn up about once in 10^4 tries)
                    u64 est = ((u64)(3 * n_add / 2 + 12) << 24) / (u64)max(n_w - n_active, 1);
                    T = (uint32_t)min(est, (u64)(1u << 24));
                }
                // Each try stages every winner with priority < T (four 64-winner chunks per pass: the list sits in
                // LDS and the four hash chains are independent), then drops the ones the segment already has:
                // the staged winners sit in registers, up to four per lane, and the kept synapses are broadcast
                // from LDS one after the other.  An empty row (a new or recycled segment, every growing segment
                // of the first steps) has nothing to drop.
                int found = 0;
                LCOUNT(9, 1);
                for (int iter = 0; iter < 64; ++iter) {
                    LCOUNT(10, 1);
                    LSTAMP(2);
                    int staged = 0;
                    for (int b0 = 0; b0 < n_w; b0 += 4 * 64) {
                        uint32_t pr[4];
                        bool take[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int i = b0 + j * 64 + lane;
                            const int wcell = i < n_w ? winner_at(i) : 0;
                            pr[j] = htm_draw24(base2, gid, enc_to_flat(d, wcell));         // :120
                            take[j] = i < n_w && pr[j] < T;
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const u64 mt = __ballot(take[j]);
                            if (take[j]) {
                                const int pos = staged + __popcll(mt & lanemask_lt());
                                if (pos < CAND_CAP) s_cand[wv][pos] = ((u64)pr[j] << 32) | (uint32_t)(b0 + j * 64 + lane);
                            }
                            staged += __popcll(mt);
                        }
                    }
                    if (staged > CAND_CAP) {                      // too many for the staging area: lower T
                        hi = T;
                        if (hi - lo <= 1) { atomicOr(&c->error, 4); found = 0; break; }
                        T = (lo + hi) / 2;
                        continue;
                    }
                    __builtin_amdgcn_wave_barrier();
                    LSTAMP(3);
                    LCOUNT(11, staged);
                    found = staged;
                    if (n_keep > 0) {                             // :121-123
                        u64 key[4];
                        int cell[4];
                        bool present[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int e = j * 64 + lane;
                            key[j] = e < staged ? s_cand[wv][e] : 0ull;
                            cell[j] = e < staged ? winner_at((int)(uint32_t)key[j]) : -2;
                            present[j] = false;
                        }
#pragma unroll 4
                        for (int f = 0; f < n_keep; ++f) {
                            const int kf = s_keep[wv][f];         // (one address for the wave: a broadcast read)
#pragma unroll
                            for (int j = 0; j < 4; ++j) present[j] |= cell[j] == kf;
                        }
                        __builtin_amdgcn_wave_barrier();          // every lane holds its entries: compact in place
                        found = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool absent = j * 64 + lane < staged && !present[j];
                            const u64 ma = __ballot(absent);
                            if (absent) s_cand[wv][found + __popcll(ma & lanemask_lt())] = key[j];
                            found += __popcll(ma);
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                    LSTAMP(4);
                    if (found >= n_add || T == (1u << 24)) break; // enough, or fewer absent winners than n_add: take all
                    lo = T;
                    T = (hi == (1u << 24)) ? (uint32_t)min((u64)T * 4u + 16u, (u64)hi) : (lo + hi + 1) / 2;
                }
                // the n_add smallest (priority, position) keys: each lane ranks its up to four candidates against the
                // list (broadcast reads again).  Only the chosen SET matters -- slots have no meaning, the reference
                // fills its free slots with the chosen cells in ascending order (:129-160) -- so the rank serves as slot
                const int n_c = min(found, CAND_CAP), take_n = min(n_add, n_c);        // :125-127
                auto rank_and_write = [&](auto slots_tag) {
                    constexpr int SL = decltype(slots_tag)::value;      // register slots in use: ceil(n_c / 64)
                    u64 key[SL];
                    int rank[SL];
#pragma unroll
                    for (int j = 0; j < SL; ++j) {
                        key[j] = j * 64 + lane < n_c ? s_cand[wv][j * 64 + lane] : ~0ull;
                        rank[j] = 0;
                    }
#pragma unroll 4
                    for (int f = 0; f < n_c; ++f) {
                        const u64 kf = s_cand[wv][f];
#pragma unroll
                        for (int j = 0; j < SL; ++j) rank[j] += kf < key[j];
                    }
#pragma unroll
                    for (int j = 0; j < SL; ++j) {
                        bool a_now = false;
                        if (j * 64 + lane < n_c && rank[j] < take_n) {
                            const int slot = n_keep + rank[j];
                            if (slot < d.E) {
                                const int wc = winner_at((int)(uint32_t)key[j]);
                                store_through(&prow[slot], wc | (d.perm_init >= d.perm_thr ? (int)SYN_CONNECTED : 0));
                                store_through(&mrow[slot], d.perm_init);                // :149,158
                                a_now = SELF && ((act_cur[wc >> 5] >> (wc & 31)) & 1u);
                            } else {
                                atomicOr(&c->error, 2);
                            }
                        }
                        if (SELF) {
                            const int na = __popcll(__ballot(a_now));
                            pot += na;
                            conn += d.perm_init >= d.perm_thr ? na : 0;
                        }
                    }
                };
                if (n_c <= 64) rank_and_write(std::integral_constant<int, 1>());
                else if (n_c <= 128) rank_and_write(std::integral_constant<int, 2>());
                else rank_and_write(std::integral_constant<int, 4>());
                n_total = min(n_keep + take_n, d.E);                                    // :161
                LSTAMP(5);
            }
        }
        if (SELF && lane == 0 && pot >= d.match_thr) {                          // role_scan's publication (:247-251, :229-239)
            const bool active = conn >= d.act_thr;
            const int cell = owner;
            const float jit = htm_jitter((float)pot, htm_draw24(base3, gid, 0u));
            atomicMax(&d.cellmax[p][cell], __float_as_uint(jit));
            if (active) atomicOr(&d.pred[p][cell >> 5], 1u << (cell & 31));
            d.seg_info[seg] = (uint32_t)pot | ((uint32_t)conn << 12) | 0x40000000u | (active ? 0x80000000u : 0u);
            d.seg_jit[seg] = jit;
            atomicOr(&d.match_bits[p][seg >> 5], 1u << (seg & 31));
        }
        // The new count clears SEG_BUSY: from then on the scan of this launch may take the row -- so the row must be there
        // first (every store of this wave acknowledged), not just issued first.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            store_through(&d.seg_nsyn[seg], n_total);
            // recyclable segments (fewer synapses than the matching threshold, projections.py:80) per 1024 ids.
            // Sharded: a death is reported with the next exchange and every rank, this one included, applies it
            // then; what becomes of a row bound this step every rank knew when it was bound (shard_bind).
            const bool was_dead = n < d.match_thr, is_dead = n_total < d.match_thr;
            if (d.world == 1) {
                if (item < n_front && was_dead != is_dead) recyc_add(d, seg >> 10, is_dead ? 1 : -1);   // (bound this step: settled at binding)
            } else if (!was_dead && is_dead) {
                const int slot = atomicAdd(&d.dead_list[0], 1);
                if (slot < DEAD_CAP) d.dead_list[1 + slot] = (int)gid; else atomicOr(&c->error, 8);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#ifdef BITHTM_LEARN_STAMPS
    if (lane == 0 && d.trace)
        for (int i = 0; i < 12; ++i) d.trace[(size_t)(blk * (BS / 64) + wv) * 16 + i] += acc_[i];
#endif
}

extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];

template <int EPL>
__global__ __launch_bounds__(RB) void k_tm_learn(Dev d, int p) {
    // (every form of a step's learning launch resets the fan-in counters its k_act_mid_rows may have used: htm_pipeline.h)
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x < FAN_COUNTERS) d.fan[(size_t)(p * FAN_COUNTERS + (int)threadIdx.x) * FAN_STRIDE] = 0u;
    role_learn<EPL, RB>(d, p, blockIdx.x, gridDim.x, (LearnShared<EPL, RB> *)dyn_lds);
}

// PredictiveProjection.process (projections.py:245-255): per segment, potential = active presynaptic cells;
// matching segments additionally count connected active synapses; per-cell prediction and max jittered
// potential (:229-239).  The last duty of a timestep: publish the next step index.
//
// Shape: 8 lanes per segment, 16-byte loads of the packed row (one 128-byte chunk = 32 synapse slots; growth tops
// a segment up to 32 active synapses, so rows rarely exceed one chunk), two segments in flight per lane group.
// A wave owns 16 consecutive segment ids per iteration, a 256-thread block 64; nothing is shared between the
// waves of a block after the bitmap of active columns has been staged in LDS, so the loop has no barrier.
//
// Per synapse the scan does as little as it can: the column bit is looked up in LDS and packed into an 8-bit hit
// mask per lane (two shifts, an and, one LDS read, a bit-field extract, a shift-or).  About 2 % of the synapses
// hit an active column; only those go on to read that column's cell word, in a loop that handles one hit per lane
// and pass -- one or two passes per wave instead of one guarded gather per slot -- and accumulates the potential
// and, from the connected flag kept in the id's top bit (`permanence >= threshold`, maintained by the learning
// role where permanences change: no permanence loads, no bit arrays), the connected-active count.
//
// Per segment the scan leaves ONE BIT (matching or not: d.match_bits, 16 bits per wave and iteration) and, for the
// few matching segments only, the info word and the jittered potential: a result word per segment cost 15-20 % of
// the stream on large pools.  The potentials of the other segments (PredictiveProjection.State.segment_potential,
// projections.py:246) are recomputed when somebody reads them (k_tm_potentials).  Recyclable segments are not
// counted here: the counts per 1024 ids are kept up to date where a row changes (role_learn, tm_bind_segment).
// LARGE: pools that are bandwidth-bound: the owner cell is fetched for matching segments only (small pools are
// latency-bound and fetch it with the synapse count, a dependent round trip earlier).
//
// Measured alternatives: a global gather for every synapse moves 64 B per bit; an LDS-only lookup (bitmap +
// prefix counts + active words) costs three bank-conflicted LDS reads per synapse and was 1.6x slower; one
// guarded cell-word gather per slot (sixteen per wave and iteration, most of them for a single lane) left the
// kernel issue-bound at ~400 vector instructions per wave and iteration.
// One 128-byte chunk of two rows, 8 slots per lane: e[u * 4 + qq] = slot `first + l * 4 + qq` of row u, of which
// row u has n[u] valid ones.  chunk_issue builds the lane's hit mask and issues the cell-word reads of its first
// two hits (lanes without a hit read act[0], one shared line: no branch, so the reads of several chunks and the
// row loads of the next iteration can all be in flight before anything is waited for); chunk_finish adds
// potential | connected-active << 16  of the lane's slots to acc[u] and loops over third and later hits (rare).
struct ChunkHits { uint32_t m_rest, e1, e2, aw1, aw2; int j1, j2; };

__device__ __forceinline__ uint32_t select8(const uint32_t (&e)[8], int j) {
    const uint32_t t0 = (j & 1) ? e[1] : e[0], t1 = (j & 1) ? e[3] : e[2], t2 = (j & 1) ? e[5] : e[4], t3 = (j & 1) ? e[7] : e[6];
    const uint32_t u0 = (j & 2) ? t1 : t0, u1 = (j & 2) ? t3 : t2;
    return (j & 4) ? u1 : u0;
}

// bit i: slot i of this lane (four of each of its two rows) is valid and its column is active (all valid ones without the
// bitmap); rows are packed: slots [0, n) are the valid ones
// (the bitmap of the step's active columns in LDS, and the bits of a cell id below its column: 5, or 6 with 64 cell slots)
struct ColBits { const uint32_t *w; int lk; };

template <bool use_lds>
__device__ __forceinline__ uint32_t chunk_mask_all(const ColBits &cb, const uint32_t (&e)[8], int first, int l, const int (&n)[2]) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t on = 1u;
        if (use_lds) {
            const uint32_t w = cb.w[(e[i] & SYN_CELL) >> (cb.lk + 5)];
            on = (w >> ((e[i] >> cb.lk) & 31)) & 1u;
        }
        m |= on << i;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int nv = min(max(n[u] - first - l * 4, 0), 4);
        m &= ~(((0xFu << nv) & 0xFu) << (4 * u));
    }
    return m;
}

__device__ __forceinline__ ChunkHits chunk_read(const uint32_t *__restrict__ act, const uint32_t (&e)[8], uint32_t m) {
    ChunkHits h;
    h.j1 = __ffs(m) - 1;                             // (-1 in lanes without a hit)
    const uint32_t m1 = m & (m - 1);
    h.j2 = __ffs(m1) - 1;
    h.m_rest = m1 & (m1 - 1);
    h.e1 = m ? select8(e, h.j1) : 0u;                // (id 0: column 0's word, bit 0 -- masked in chunk_finish)
    h.e2 = m1 ? select8(e, h.j2) : 0u;
    h.aw1 = act[(h.e1 & SYN_CELL) >> 5];
    h.aw2 = act[(h.e2 & SYN_CELL) >> 5];
    return h;
}

template <bool use_lds>
__device__ __forceinline__ ChunkHits chunk_issue(const uint32_t *__restrict__ act, const ColBits &s_colbits, const uint32_t (&e)[8],
                                                 int first, int l, const int (&n)[2]) {
    return chunk_read(act, e, chunk_mask_all<use_lds>(s_colbits, e, first, l, n));
}

template <bool BATCH>
__device__ __forceinline__ void chunk_finish(const uint32_t *__restrict__ act, const uint32_t (&e)[8], const ChunkHits &h, uint32_t (&acc)[2]) {
    {
        const uint32_t a = h.j1 >= 0 ? (h.aw1 >> (h.e1 & 31)) & 1u : 0u;
        const uint32_t add = a + ((a & (h.e1 >> 31)) << 16);
        acc[0] += (h.j1 & 4) ? 0u : add;
        acc[1] += (h.j1 & 4) ? add : 0u;
    }
    {
        const uint32_t a = h.j2 >= 0 ? (h.aw2 >> (h.e2 & 31)) & 1u : 0u;
        const uint32_t add = a + ((a & (h.e2 >> 31)) << 16);
        acc[0] += (h.j2 & 4) ? 0u : add;
        acc[1] += (h.j2 & 4) ? add : 0u;
    }
    // Third and later hits of a lane.  Rare among random synapses (4 lanes in 10 000) -- but the synapses of a MATCHING
    // segment are mostly active, and segments created together match together: whole blocks of them.  Their reads
    // are therefore all issued before any is used (one pass per hit, with a dependent read each, made those blocks
    // the tail of the launch); the passes beyond the lanes' largest hit count are skipped by wave-uniform branches.
    // (BATCH = false, the large-pool kernels: one hit per pass -- the extra registers cost them a wave per SIMD)
    if (!BATCH) {
        uint32_t m = h.m_rest;
        while (__any(m != 0)) {
            const int j = __ffs(m) - 1;
            const uint32_t ej = select8(e, j);
            uint32_t aw = 0;
            if (m) aw = act[(ej & SYN_CELL) >> 5];
            const uint32_t a = (aw >> (ej & 31)) & 1u;   // (aw = 0 without a hit)
            const uint32_t add = a + ((a & (ej >> 31)) << 16);
            acc[0] += (j & 4) ? 0u : add;
            acc[1] += (j & 4) ? add : 0u;
            m &= m - 1;
        }
        return;
    }
    uint32_t rest = h.m_rest;
    if (__any(rest != 0)) {                          // hits three to eight of a lane: all reads first
        // (a lane holds four slots of each of two rows, and nearly every synapse of a matching segment is active: the
        // lanes of two matching neighbours have eight hits each -- with the last two read one dependent pass after the
        // other, those blocks took 9 us against the others' 5)
        uint32_t aw[6];
        uint32_t m = rest;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            aw[t] = 0;
            if (t > 0 && !__any(m != 0)) break;
            const uint32_t ej = m ? select8(e, __ffs(m) - 1) : 0u;
            aw[t] = act[(ej & SYN_CELL) >> 5];
            m &= m - 1;
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            if (t > 0 && !__any(rest != 0)) break;
            const int j = __ffs(rest) - 1;
            const uint32_t ej = rest ? select8(e, j) : 0u;
            const uint32_t a = rest ? (aw[t] >> (ej & 31)) & 1u : 0u;
            const uint32_t add = a + ((a & (ej >> 31)) << 16);
            acc[0] += (j & 4) ? 0u : add;
            acc[1] += (j & 4) ? add : 0u;
            rest &= rest - 1;
        }
    }
}

// The same for small pools (latency-bound: every block is resident at once and the launch is as long as its slowest
// block): ALL of a lane's cell-word reads are issued here, the first two unconditionally and the others while any lane
// of the wave still has a hit -- a lane holds four slots of each of two rows and nearly every synapse of a matching
// segment is active, so the lanes of two matching neighbours have eight hits each; read in dependent passes (two, then
// four, then one by one) those blocks took 9 us against the others' 5.
struct ChunkHitsAll { uint32_t m; uint32_t aw[2]; };

template <int NOW>
__device__ __forceinline__ ChunkHitsAll chunk_read_all(const uint32_t *__restrict__ act, const uint32_t (&e)[8], uint32_t m) {
    ChunkHitsAll h;
    h.m = m;
    uint32_t mm = m;
#pragma unroll
    for (int t = 0; t < NOW; ++t) {
        const uint32_t ej = mm ? select8(e, __ffs(mm) - 1) : 0u;     // (lanes without a hit read entry 0: no branch)
        h.aw[t] = act[(ej & SYN_CELL) >> 5];
        mm &= mm - 1;
    }
    return h;
}

template <bool use_lds, int NOW>
__device__ __forceinline__ ChunkHitsAll chunk_issue_all(const uint32_t *__restrict__ act, const ColBits &s_colbits, const uint32_t (&e)[8],
                                                        int first, int l, const int (&n)[2]) {
    return chunk_read_all<NOW>(act, e, chunk_mask_all<use_lds>(s_colbits, e, first, l, n));
}

// NOW = what chunk_issue_all has read; the rest is read here, three hits per pass (all three reads before any is used)
template <int NOW>
__device__ __forceinline__ void chunk_finish_all(const uint32_t *__restrict__ act, const uint32_t (&e)[8], const ChunkHitsAll &h, uint32_t (&acc)[2]) {
    auto count = [&](uint32_t mm, uint32_t aw) {     // the lane's lowest remaining hit, if it has one
        const int j = __ffs(mm) - 1;
        const uint32_t ej = mm ? select8(e, j) : 0u;
        const uint32_t a = mm ? (aw >> (ej & 31)) & 1u : 0u;
        const uint32_t add = a + ((a & (ej >> 31)) << 16);
        acc[0] += (j & 4) ? 0u : add;
        acc[1] += (j & 4) ? add : 0u;
    };
    uint32_t mm = h.m;
#pragma unroll
    for (int t = 0; t < NOW; ++t) {
        count(mm, h.aw[t]);
        mm &= mm - 1;
    }
    while (__any(mm != 0)) {
        uint32_t aw[3], m2 = mm;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const uint32_t ej = m2 ? select8(e, __ffs(m2) - 1) : 0u;
            aw[t] = act[(ej & SYN_CELL) >> 5];
            m2 &= m2 - 1;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            count(mm, aw[t]);
            mm &= mm - 1;
        }
    }
}

// The same out of LDS (three-launch schedule): the active word of an ACTIVE column is act_list[its position in the
// ascending winner list], and that position is col_rank[bitmap word] + the set bits below the column's in the word --
// both tables written by the step's select finish and activation, staged beside the bitmap.  Three LDS reads per
// slot instead of a divergent global read of 4 bytes in 64 different lines per instruction: a group of matching
// segments (nearly every synapse a hit, 16 per lane over two chunks) took 6-8 us that way, the tail of the launch.
// Every slot of the lane is looked up (no hit-by-hit passes: the reads are cheap), slots outside m count nothing.
// (lk: bits of a cell id below its column; with 64 cell slots an active column has two active words in the table, side by side)
struct ScanTabs { const uint32_t *colbits; const uint16_t *rank; const uint32_t *actw; int lk; };

__device__ __forceinline__ void chunk_count_tab(const ScanTabs &tb, const uint32_t (&e)[8], uint32_t m, uint32_t (&acc)[2]) {
    uint32_t aw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t col = (e[i] & SYN_CELL) >> tb.lk, cw = col >> 5;
        const uint32_t w = tb.colbits[cw];
        uint32_t rk = (uint32_t)tb.rank[cw] + (uint32_t)__popc(w & ((1u << (col & 31)) - 1u));
        if (tb.lk == 6) rk = 2u * rk + ((e[i] >> 5) & 1u);
        const uint32_t on = (m >> i) & 1u;                 // (a slot outside m may name any column: entry 0 is read for it,
        aw[i] = tb.actw[on ? rk : 0u] & (0u - on);         //  no branch -- a guarded read is waited for slot by slot)
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t a = (aw[i] >> (e[i] & 31)) & 1u;
        acc[i >> 2] += a + ((a & (e[i] >> 31)) << 16);
    }
}

// diagnostic build (-DBITHTM_SCAN_STAMPS, handle created under BITHTM_TRACE=1): device clock at the phases of the first
// iteration of every wave of the first 2048 scan blocks, d.trace[(block * 4 + wave) * 8 + phase] (tools/scan_phases.py)
#ifdef BITHTM_SCAN_STAMPS
// (-DBITHTM_SCAN_STAMPS=2: the phases of the wave's LAST iteration instead, slot 1 = when that iteration began)
#define SCAN_STAMP(i) do { if (d.trace && blk < 2048 && BS == 256 && (threadIdx.x & 63) == 0 && (BITHTM_SCAN_STAMPS == 2 ? !first_iter : first_iter)) d.trace[(size_t)(blk * 4 + wave) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define SCAN_STAMP(i) do { } while (0)
#endif

// LDS: from word 4: column bitmap [colwords] (words 0..3 unused)
// TAB (small pools under the LDS bitmap, three-launch schedule): + rank [colwords] u16, active words [k + 8]
// DYN (the streaming form inside the three-launch schedule's last launch): EVERY block of that launch scans.  The launch's other
// blocks -- the select finish, the learning role -- are done after 7 to 16 us of its ~50; with scan blocks only, the ones
// dispatched into their slots end that much later than the rest, or, with more blocks than slots, in a second round that
// leaves the chip half empty (2 048 blocks over 1 280 slots: the last block ended 13 us after the chip had begun to
// drain).  Here the grid is what is resident at once and a block whose other role is done JOINS the scan (blk < 0; a barrier
// first: its LDS is still in use by slower waves of the other role).  The groups of 16 segments are dealt in rounds: in
// every round each wave of the launch takes one group -- equal shares.  (Unequal ones were tried: a wave of the select finish's
// blocks, dispatched first and therefore served first by SIMDs that issue oldest-wave-first, takes 1.5 us per group, one of
// the scan blocks, dispatched last, 2.6 -- but shares of 6/8/7 or 5/8/6 rounds in eight for scan / select finish / learning
// blocks all measured within a microsecond of equal ones: the launch is bound by what the chip issues and fetches in total,
// and whoever is left runs faster once the others are done.  And handing the groups out on demand instead -- one returning
// atomic per wave and group, on 64 counters a page apart -- made the launch five times LONGER: the chip completes about 300
// such atomics per microsecond, whatever their addresses.)
struct ScanRounds {
    int n_waves, idx;                              // scanning waves of the launch (every block's), this wave's index among them
    __device__ __forceinline__ int group(int r) const { return r * n_waves + idx; }
};

template <int BS, bool use_lds, bool LARGE, bool TAB = false, bool DYN = false>
__device__ __forceinline__ void role_scan(const Dev &d, int p, int blk, int nblk, int n_spec, uint32_t *lds) {
    static_assert(BS % 64 == 0, "whole waves of 16 segments");
    static_assert(!TAB || use_lds, "the tables go with the LDS bitmap");
    static_assert(!DYN || LARGE, "joining blocks belong to the streaming form");
    static_assert(!(LARGE && TAB), "the streaming form reads the cell words from memory");
    if (DYN && blk < 0) __syncthreads();
    constexpr int U = 2;                           // segments in flight per lane group
    // QUEUE (the streaming form in 256-thread blocks): rows that need more than their first chunk are set aside and counted
    // sixteen at a time.  A learned pool is not the pre-populated one: 46 % of the 350-pattern pool's rows hold 33-64 synapses,
    // 7 % more than 47 (which can match whatever their first chunk says), and about one row in ten goes on to its second
    // chunk -- one such row among a wave's sixteen, four waves in five, and the WHOLE wave masked, looked up and gathered a
    // second chunk: the scan's VALU and LDS work nearly doubled (56 us for 1.57 M learned rows against 38 for as many
    // one-chunk rows).  Now such a row leaves its first chunk's counts in a queue of the wave (LDS, 32 entries), and when
    // sixteen are waiting the wave counts their further chunks in one dense pass (and the rest when it runs out of rows).
    constexpr bool QUEUE = LARGE && BS == 256;
    constexpr int QCAP = 32;
    uint32_t *s_colbits = lds + 4;
    const int rank_q = (d.colwords * 2 + 15) / 16, actw_q = (d.k * d.WPC + 8 + 3) / 4;      // 16-byte units of the two tables
    uint16_t *s_rank = (uint16_t *)(s_colbits + ((d.colwords + 3) & ~3));
    uint32_t *s_actw = (uint32_t *)s_rank + rank_q * 4;
    const ScanTabs tabs{s_colbits, s_rank, s_actw, d.LK};
    (void)tabs;
    const ColBits cbits{s_colbits, d.LK};
    uint32_t *s_queue = lds + 4 + (use_lds ? ((d.colwords + 3) & ~3) : 0) + (threadIdx.x >> 6) * (QCAP * 3);      // (this wave's: row, synapses, counts so far)
    int qn = 0;                                      // rows waiting in it (wave-uniform)
    constexpr bool need_cell = !LARGE;
    Counters *c = d.ctr;
    const int S = d.world > 1 ? c->L : c->S;         // rows to scan (a shard scans its local rows; a free row is empty)
    if (blk == 0 && threadIdx.x == 0) {
        c->step[p ^ 1] = c->step[p] + 1;
        c->has_distal = 1;
        c->n_work[p ^ 1] = 0;                       // (the coming step's; this step's learning role may still be reading its own)
        c->n_bind[p ^ 1] = 0;
    }
    const uint32_t *act = d.act[p];
    const uint32_t base3 = htm_stream_base(d.seed, HTM_STREAM_SEGMENT_JITTER, c->step[p]);
    const int wave = threadIdx.x >> 6, gi = (threadIdx.x & 63) >> 3, l = threadIdx.x & 7;
    // Which 16 segments a wave takes.  Large pools stream: block after block, wave after wave.  Small pools are
    // latency-bound and every block is resident at once: there the waves of a block take groups 256 blocks apart --
    // segments created together are used together (a pattern's ~1 300 segments are 80 consecutive groups, nearly all of
    // their synapses active at once), and four such waves on one CU, each gathering a thousand cell words, were the tail
    // of the launch (8.2 us against 4.6 for the others); one per CU is not.
    constexpr int WPB = BS / 64;
    const int sg = LARGE ? 1 : (nblk % 256 == 0 ? 256 : nblk);
    const int gw = DYN ? 0 : LARGE ? wave : wave * sg;                                // this wave's group within the block's
    // (DYN: nblk = scan blocks of the launch, n_spec = the blocks that join it after another role; a joiner's blk = -1 - its index)
    const ScanRounds rounds{(nblk + n_spec) * WPB, (blk >= 0 ? blk : nblk + (-1 - blk)) * WPB + wave};
    int round = 0;
    const int g_first = DYN ? rounds.group(0) : LARGE ? blk * WPB : (blk / sg) * (sg * WPB) + blk % sg;      // the block's first group (DYN: the wave's)
    const int gstride = nblk * WPB;
    // round trip 1 of an iteration: synapse count, owner cell and the first chunk of each row, all unconditional
    // (rows are clamped to the pool and masked once the row count is known)
    struct Batch { int seg[U], n[U], cell[U]; int4 ps[U]; };
    auto fetch = [&](int b) {
        Batch t;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            t.seg[u] = min((b + gw) * 16 + u * 8 + gi, d.Lcap - 1);
            t.n[u] = d.seg_nsyn[t.seg[u]];
            t.cell[u] = need_cell ? d.seg_cell[t.seg[u]] : 0;
            t.ps[u] = *(const int4 *)(d.presyn + (size_t)t.seg[u] * d.E + l * 4);
        }
        return t;
    };
    // In the first n_spec blocks (the ones that had segments when the host last saw the segment count)
    // the loads of the first batch do not wait for the count: one dependent round trip less.
    const bool speculative = !DYN && g_first * 16 < n_spec * SCAN_SEGS;
    if (!DYN && !speculative && g_first * 16 >= S) return;       // (DYN: a wave without a group still meets the block's barrier)
    bool first_iter = true;
    (void)first_iter;
#ifdef BITHTM_SCAN_STAMPS
    int n_iter = 1;
#endif
    SCAN_STAMP(0);
    Batch cur = fetch(g_first);
    if (use_lds)                                     // the bitmap staging overlaps with those loads
        for (int i = threadIdx.x; i < d.colwords; i += BS) s_colbits[i] = d.colbits[p][i];
    if (TAB) {                                       // (16 bytes per load, no guard: both arrays are allocated in whole units and zero beyond their end)
        const uint4 *rk = (const uint4 *)d.col_rank[p], *aw = (const uint4 *)d.act_list;
        for (int i = threadIdx.x; i < rank_q; i += BS) ((uint4 *)s_rank)[i] = rk[i];
        for (int i = threadIdx.x; i < actw_q; i += BS) ((uint4 *)s_actw)[i] = aw[i];
    }
    __syncthreads();                                 // the only barrier: from here on the waves share nothing
    SCAN_STAMP(1);
    // QUEUE: the last `count` (<= 16) rows waiting in the wave's queue, their further chunks and their publication
    auto flush = [&](int count) __attribute__((always_inline)) {
        const int qbase = qn - count;
        int qs[U], qlen[U];
        uint32_t qacc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = u * 8 + gi;
            const bool ok = r < count;
            const uint32_t *ent = s_queue + (qbase + (ok ? r : 0)) * 3;
            qs[u] = ok ? (int)ent[0] : S;
            qlen[u] = ok ? (int)ent[1] : 0;
            qacc[u] = (ok && l == 0) ? ent[2] : 0u;      // (the first chunk's counts, already summed over the row's lanes)
        }
        for (int cc = 1; __any(qlen[0] > cc * 32 || qlen[1] > cc * 32); ++cc) {
            int4 pa[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                pa[u] = qlen[u] > cc * 32 ? *(const int4 *)(d.presyn + (size_t)qs[u] * d.E + cc * 32 + l * 4) : make_int4(0, 0, 0, 0);
            const uint32_t ea[8] = {(uint32_t)pa[0].x, (uint32_t)pa[0].y, (uint32_t)pa[0].z, (uint32_t)pa[0].w,
                                    (uint32_t)pa[1].x, (uint32_t)pa[1].y, (uint32_t)pa[1].z, (uint32_t)pa[1].w};
            const ChunkHits ha = chunk_issue<use_lds>(act, cbits, ea, cc * 32, l, qlen);
            chunk_finish<false>(act, ea, ha, qacc);
        }
        bool mt[U];
        int qpot[U], qconn[U], qcell[U];
        float qjit[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t sum = (uint32_t)group8_sum_first((int)qacc[u]);
            qpot[u] = (int)(sum & 0xFFFFu);
            qconn[u] = (int)(sum >> 16);
            mt[u] = l == 0 && qs[u] < S && qpot[u] >= d.match_thr;                  // :247
            qjit[u] = 0.f;
            qcell[u] = 0;
        }
        if (__any(mt[0] || mt[1])) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t gid = mt[u] ? (uint32_t)seg_gid_of(d, qs[u]) : 0u;
                qcell[u] = mt[u] ? d.seg_cell[qs[u]] : 0;
                qjit[u] = htm_jitter((float)qpot[u], htm_draw24(base3, gid, 0u));   // :234-235
            }
            asm volatile("" : "+v"(qjit[0]), "+v"(qjit[1]), "+v"(qcell[0]), "+v"(qcell[1]) : : "memory");
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (mt[u]) {
                    const bool active = qconn[u] >= d.act_thr;                       // :250
                    atomicMax(&d.cellmax[p][qcell[u]], __float_as_uint(qjit[u]));   // :237
                    if (active) atomicOr(&d.pred[p][qcell[u] >> 5], 1u << (qcell[u] & 31));
                    d.seg_info[qs[u]] = (uint32_t)qpot[u] | ((uint32_t)qconn[u] << 12) | 0x40000000u | (active ? 0x80000000u : 0u);
                    d.seg_jit[qs[u]] = qjit[u];
                    atomicOr(&d.match_bits[p][qs[u] >> 5], 1u << (qs[u] & 31));
                }
        }
        qn -= count;
    };
    for (int b = g_first; b * 16 < S; b += gstride) {    // b = the block's first group of this iteration
        int seg[U], n[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = (b + gw) * 16 + u * 8 + gi < S;
            n[u] = (ok && !(cur.n[u] & (int)SEG_BUSY)) ? cur.n[u] : 0;       // (a row the learning role is rewriting is its to scan)
            seg[u] = ok ? cur.seg[u] : S;
        }
        uint32_t acc[U] = {0u, 0u};                  // this lane's share of potential (:247) | connected-active count << 16 (:171-172)
        const uint32_t e1[8] = {(uint32_t)cur.ps[0].x, (uint32_t)cur.ps[0].y, (uint32_t)cur.ps[0].z, (uint32_t)cur.ps[0].w,
                                (uint32_t)cur.ps[1].x, (uint32_t)cur.ps[1].y, (uint32_t)cur.ps[1].z, (uint32_t)cur.ps[1].w};
        // A row is looked up in the cell words only if it can match at all, and its later chunks are read only then:
        // its synapses in ACTIVE COLUMNS among the first 32 (known from the bitmap in LDS, before any cell word is read)
        // plus all of its later synapses must reach the matching threshold (:247).  A row that cannot has a potential
        // below the threshold whatever its cells do, and nothing else of it is published: its hits are dropped from the
        // mask.  All but the rows of the patterns that are showing go that way -- their divergent cell-word reads
        // were most of a wave's instructions, and the second lines of the 35 % of rows longer than a chunk a fifth of
        // the launch's traffic.
        uint32_t m1 = chunk_mask_all<use_lds>(cbits, e1, 0, l, n);
        if (use_lds) {
            const int hits = group8_sum_all((int)(__popc(m1 & 0xFu) | (__popc(m1 >> 4) << 8)));
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (((hits >> (8 * u)) & 0xFF) + max(n[u] - 32, 0) < d.match_thr) {
                    n[u] = min(n[u], 32);
                    m1 &= ~(0xFu << (4 * u));
                }
        }
        // QUEUE: a row that goes on past its first chunk (it can match, and is longer) is set aside below; here it ends at 32
        const int n_full[U] = {n[0], n[1]};
        if (QUEUE) { n[0] = min(n[0], 32); n[1] = min(n[1], 32); }
        // round trip 2 (only rows longer than one chunk): second chunk, in flight during the lookups of the first
        int4 ps2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ps2[u] = make_int4(0, 0, 0, 0);
            if (n[u] > 32) ps2[u] = *(const int4 *)(d.presyn + (size_t)seg[u] * d.E + 32 + l * 4);
        }
        SCAN_STAMP(2);                               // (synapse counts are here)
        ChunkHits h1;
        ChunkHitsAll g1;
        const bool any_hit = __any(m1 != 0);
        g1.m = 0; g1.aw[0] = g1.aw[1] = 0;
        if (LARGE && !TAB) h1 = chunk_read(act, e1, m1);
        else if (!TAB && any_hit) g1 = chunk_read_all<2>(act, e1, m1);
        // large pools: the next iteration's rows are requested now, behind this iteration's cell-word reads (loads
        // return in issue order: requested earlier they would be waited for with those reads)
        if (DYN) ++round;
        const int b_next = DYN ? rounds.group(round) : b + gstride;
        const int cell_cur[U] = {cur.cell[0], cur.cell[1]};
        // (unconditional -- past the last batch the clamped ids fetch a row nobody uses: a branch around the loads
        // would make the compiler wait for them with everything else)
        Batch nxt = cur;
        if (LARGE) nxt = fetch(b_next);

        const bool any_long = __any(n[0] > 32 || n[1] > 32);         // (a row that long which can still match: few waves have one)
        const uint32_t e2[8] = {(uint32_t)ps2[0].x, (uint32_t)ps2[0].y, (uint32_t)ps2[0].z, (uint32_t)ps2[0].w,
                                (uint32_t)ps2[1].x, (uint32_t)ps2[1].y, (uint32_t)ps2[1].z, (uint32_t)ps2[1].w};
        if (TAB) {
            if (any_hit) chunk_count_tab(tabs, e1, m1, acc);
            if (any_long) chunk_count_tab(tabs, e2, chunk_mask_all<use_lds>(cbits, e2, 32, l, n), acc);
        } else if (!LARGE) {
            // small pools: the second chunk's first cell-word reads go out before the first chunk's are waited for
            // (its rows were requested before the first chunk's lookups): one round trip less on the blocks' chain
            ChunkHitsAll g2 = g1;
            if (any_long) g2 = chunk_issue_all<use_lds, 2>(act, cbits, e2, 32, l, n);
            if (any_hit) chunk_finish_all<2>(act, e1, g1, acc);      // (most waves have no row that can match)
            if (any_long) chunk_finish_all<2>(act, e2, g2, acc);
        } else {
            chunk_finish<false>(act, e1, h1, acc);
            if (any_long) {                          // (skipped by waves in which no row is that long)
                const ChunkHits h2 = chunk_issue<use_lds>(act, cbits, e2, 32, l, n);
                chunk_finish<false>(act, e2, h2, acc);
            }
        }
        // rows longer than two chunks (a few per thousand at cfg 3, and what set the kernel's tail: five dependent round
        // trips per further chunk when they were read and looked up one after the other): two chunks per pass, their
        // four row loads in flight together, then their cell-word reads
        // (the large-pool kernels take one chunk per pass: the second pair of rows in registers costs them a wave per SIMD)
        for (int c = 2; __any(n[0] > c * 32 || n[1] > c * 32); c += (LARGE && !TAB) ? 1 : 2) {
            int4 pa[U], pb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int *prow = d.presyn + (size_t)seg[u] * d.E + l * 4;
                pa[u] = n[u] > c * 32 ? *(const int4 *)(prow + c * 32) : make_int4(0, 0, 0, 0);
                if (!LARGE || TAB) pb[u] = n[u] > (c + 1) * 32 ? *(const int4 *)(prow + (c + 1) * 32) : make_int4(0, 0, 0, 0);
            }
            const uint32_t ea[8] = {(uint32_t)pa[0].x, (uint32_t)pa[0].y, (uint32_t)pa[0].z, (uint32_t)pa[0].w,
                                    (uint32_t)pa[1].x, (uint32_t)pa[1].y, (uint32_t)pa[1].z, (uint32_t)pa[1].w};
            ChunkHits ha;
            ChunkHitsAll ga;
            if (TAB) chunk_count_tab(tabs, ea, chunk_mask_all<use_lds>(cbits, ea, c * 32, l, n), acc);
            else if (LARGE) ha = chunk_issue<use_lds>(act, cbits, ea, c * 32, l, n);
            else ga = chunk_issue_all<use_lds, 2>(act, cbits, ea, c * 32, l, n);
            if (TAB) {
                const uint32_t eb[8] = {(uint32_t)pb[0].x, (uint32_t)pb[0].y, (uint32_t)pb[0].z, (uint32_t)pb[0].w,
                                        (uint32_t)pb[1].x, (uint32_t)pb[1].y, (uint32_t)pb[1].z, (uint32_t)pb[1].w};
                chunk_count_tab(tabs, eb, chunk_mask_all<use_lds>(cbits, eb, (c + 1) * 32, l, n), acc);
            } else if (!LARGE) {
                const uint32_t eb[8] = {(uint32_t)pb[0].x, (uint32_t)pb[0].y, (uint32_t)pb[0].z, (uint32_t)pb[0].w,
                                        (uint32_t)pb[1].x, (uint32_t)pb[1].y, (uint32_t)pb[1].z, (uint32_t)pb[1].w};
                const ChunkHitsAll gb = chunk_issue_all<use_lds, 2>(act, cbits, eb, (c + 1) * 32, l, n);
                chunk_finish_all<2>(act, ea, ga, acc);
                chunk_finish_all<2>(act, eb, gb, acc);
            } else {
                chunk_finish<false>(act, ea, ha, acc);
            }
        }
        SCAN_STAMP(3);                               // (every synapse counted)
        // What the publication reads from memory (global ids on a shard, owner cells in the large-pool form) is read,
        // and everything computed from it, for both segments before either is published: behind the first segment's
        // atomics and stores the wait for such a read is a wait for them as well (the compiler cannot tell them
        // apart across the branches) -- a microsecond per matching segment pair, in the waves that have them.
        bool matching[U];
        int pot[U], conn[U], cell_of[U];
        float jit[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t sum = (uint32_t)group8_sum_first((int)acc[u]);     // (valid in the segment's first lane only)
            pot[u] = (int)(sum & 0xFFFFu);
            conn[u] = (int)(sum >> 16);
            matching[u] = l == 0 && seg[u] < S && pot[u] >= d.match_thr && !(QUEUE && n_full[u] > 32);      // :247 (a row set aside is not done)
            jit[u] = 0.f;
            cell_of[u] = 0;
        }
        if (QUEUE) {                                 // rows that go on past their first chunk: into the wave's queue
            const bool q0 = l == 0 && n_full[0] > 32, q1 = l == 0 && n_full[1] > 32;
            const u64 d0 = __ballot(q0), d1 = __ballot(q1);
            if (d0 | d1) {                           // (wave-uniform)
                if (qn > QCAP - 16) flush(16);
                const int pos = qn + (q0 ? __popcll(d0 & lanemask_lt()) : __popcll(d0) + __popcll(d1 & lanemask_lt()));
                if (q0) { s_queue[pos * 3] = (uint32_t)seg[0]; s_queue[pos * 3 + 1] = (uint32_t)n_full[0]; s_queue[pos * 3 + 2] = (uint32_t)pot[0] | ((uint32_t)conn[0] << 16); }
                if (q1) {
                    const int pos1 = qn + __popcll(d0) + __popcll(d1 & lanemask_lt());
                    s_queue[pos1 * 3] = (uint32_t)seg[1]; s_queue[pos1 * 3 + 1] = (uint32_t)n_full[1]; s_queue[pos1 * 3 + 2] = (uint32_t)pot[1] | ((uint32_t)conn[1] << 16);
                }
                (void)pos;
                qn += __popcll(d0) + __popcll(d1);
            }
        }
        if (__any(matching[0] || matching[1])) {     // (few waves: the streaming forms must not pay for the arithmetic)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t gid = matching[u] ? (uint32_t)seg_gid_of(d, seg[u]) : 0u;
                cell_of[u] = need_cell ? cell_cur[u] : (matching[u] ? d.seg_cell[seg[u]] : 0);
                jit[u] = htm_jitter((float)pot[u], htm_draw24(base3, gid, 0u));       // :234-235
            }
            // (the compiler would sink the arithmetic back under the branches, and the wait with it)
            asm volatile("" : "+v"(jit[0]), "+v"(jit[1]), "+v"(cell_of[0]), "+v"(cell_of[1]) : : "memory");
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (matching[u]) {
                    const bool active = conn[u] >= d.act_thr;                         // :250
                    const int cell = cell_of[u];
                    atomicMax(&d.cellmax[p][cell], __float_as_uint(jit[u]));         // :237
                    if (active) atomicOr(&d.pred[p][cell >> 5], 1u << (cell & 31));   // :251, networks.py:122
                    d.seg_info[seg[u]] = (uint32_t)pot[u] | ((uint32_t)conn[u] << 12) | 0x40000000u | (active ? 0x80000000u : 0u);
                    d.seg_jit[seg[u]] = jit[u];
                }
        }
        SCAN_STAMP(4);                               // (matching segments published)
        {   // the wave's 16 match bits: the ballots hold one bit per lane group at lane 8 * gi; a multiplication
            // gathers those eight bits into the top byte (all partial products fall on different bit positions)
            const u64 m0 = __ballot(matching[0]), m1 = __ballot(matching[1]);
            const uint32_t b0 = (uint32_t)(((m0 & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
            const uint32_t b1 = (uint32_t)(((m1 & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
            const uint32_t bits = b0 | (b1 << 8);      // (the words were zeroed by this step's middle launch)
            if ((threadIdx.x & 63) == 0 && bits) atomicOr(&d.match_bits[p][(b + gw) >> 1], bits << (16 * ((b + gw) & 1)));
        }
        SCAN_STAMP(5);
        first_iter = false;
        if (DYN) b = b_next - gstride;                // (the loop adds the stride back)
        if (b_next * 16 >= S) break;
        cur = LARGE ? nxt : fetch(b_next);
#ifdef BITHTM_SCAN_STAMPS
        ++n_iter;
        SCAN_STAMP(1);
#endif
    }
    if (QUEUE)
        while (qn > 0) flush(min(qn, 16));           // the rows still waiting
#ifdef BITHTM_SCAN_STAMPS                            // when the wave left, and how many groups it took
    if (d.trace && blk < 2048 && BS == 256 && (threadIdx.x & 63) == 0) {
        d.trace[(size_t)(blk * 4 + wave) * 8 + 6] = wall_clock64();
        d.trace[(size_t)(blk * 4 + wave) * 8 + 7] = (unsigned long long)n_iter;
    }
#endif
}

// use_lds is a compile-time switch: as a run-time flag it put a branch and a wait around every
// single LDS lookup, which serialised them
// MINW = 6 caps the kernel at 80 registers so that 6 blocks fit a CU and a pool of up to ~98 k
// segments is scanned by blocks that are all resident at once (the latency-bound regime of the bench
// workload); large pools are bandwidth-bound and run faster without the cap (MINW = 1).
template <bool use_lds, int MINW>
__global__ __launch_bounds__(256, MINW) void k_tm_scan(Dev d, int p, int n_spec) {
    role_scan<256, use_lds, MINW == 1>(d, p, blockIdx.x, gridDim.x, n_spec, (uint32_t *)dyn_lds);
}

// the same for large pools under a big column bitmap: 1024-thread blocks (16 waves share one copy of the bitmap), two per CU
template <bool use_lds>
__global__ __launch_bounds__(1024, 8) void k_tm_scan_wide(Dev d, int p, int n_spec) {
    role_scan<1024, use_lds, true>(d, p, blockIdx.x, gridDim.x, n_spec, (uint32_t *)dyn_lds);
}

// htm_populate: one wave per segment, lane i = synapse i.  Row q of the rank's own range: cell own_lo + q / spc,
// segment id (cell - cell_begin) * spc + q % spc.  Presynaptic cells: (draw32 * N) >> 32, made distinct within the
// segment in synapse order (a cell already taken by an earlier synapse moves on to the next free cell id, mod N).
__global__ __launch_bounds__(256) void k_tm_populate(Dev d, long long cell_begin, long long own_lo, long long own_rows, int spc, int n_syn,
                                                     double perm_lo, double perm_hi, uint32_t seed) {
    const int lane = lane_id();
    const uint32_t N = (uint32_t)d.C * (uint32_t)d.K;
    const uint32_t base_c = htm_stream_base(seed, HTM_STREAM_POPULATE_CELL, 0u), base_p = htm_stream_base(seed, HTM_STREAM_POPULATE_PERM, 0u);
    for (long long q = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6; q < own_rows; q += ((long long)gridDim.x * 256) >> 6) {
        const long long flat = own_lo + q / spc;
        const int j = (int)(q % spc);
        const uint32_t gid = (uint32_t)((flat - cell_begin) * spc + j);
        const int row = d.world > 1 ? (int)q : (int)gid;
        const bool on = lane < n_syn;
        uint32_t cell = on ? (uint32_t)(((u64)htm_draw32(base_c, gid, (uint32_t)lane) * N) >> 32) : 0xFFFFFFFFu;
        bool dup = false;                           // an earlier synapse drew the same cell?
        for (int t = 0; t < n_syn; ++t) {
            const uint32_t ct = (uint32_t)__builtin_amdgcn_readlane((int)cell, t);
            dup |= on && lane > t && cell == ct;
        }
        if (__any(dup)) {                           // rare (about n_syn^2 / 2N of the segments): settle them in synapse order
            for (int i = 1; i < n_syn; ++i) {
                uint32_t ci = (uint32_t)__builtin_amdgcn_readlane((int)cell, i);
                while (__any(on && lane < i && cell == ci)) ci = ci + 1u == N ? 0u : ci + 1u;
                if (lane == i) cell = ci;
            }
        }
        if (on) {
            const uint32_t u = htm_draw24(base_p, gid, (uint32_t)lane);
            const float perm = (float)(perm_lo + (perm_hi - perm_lo) * ((double)u * (1.0 / 16777216.0)));
            const int enc = (int)((cell / (uint32_t)d.K) * (uint32_t)d.KP + cell % (uint32_t)d.K);
            d.presyn[(size_t)row * d.E + lane] = enc | (perm >= d.perm_thr ? (int)SYN_CONNECTED : 0);
            d.sperm[(size_t)row * d.E + lane] = perm;
        }
        if (lane == 0) {
            const int owner = (int)((flat / d.K) * d.KP + flat % d.K);
            d.seg_cell[row] = owner;
            d.seg_nsyn[row] = n_syn;
            if (d.seg_gid) { d.seg_gid[row] = (int)gid; d.g2l[gid] = row; }
            if (j == 0) d.segcount[owner] = spc;
        }
    }
}

// State.segment_potential (projections.py:246) for every segment, on demand: active presynaptic cells of the last
// completed step (parity p), 8 lanes per segment (sharded handles: per local row).  Rows [row_begin, row_end) into
// out[row - row_begin].
__global__ __launch_bounds__(256) void k_tm_potentials(Dev d, int p, int *out, int row_begin, int row_end) {
    const int S = min(d.world > 1 ? d.ctr->L : d.ctr->S, row_end);
    const uint32_t *act = d.act[p];
    const int l = threadIdx.x & 7;
    out -= row_begin;
    for (int seg = row_begin + ((blockIdx.x * 256 + threadIdx.x) >> 3); seg < S; seg += (gridDim.x * 256) >> 3) {
        const int n = d.seg_nsyn[seg];
        const int *prow = d.presyn + (size_t)seg * d.E;
        int pot = 0;
        for (int i = l; i < n; i += 8) {
            const int e = prow[i] & SYN_CELL;
            pot += (int)((act[e >> 5] >> (e & 31)) & 1u);
        }
        pot = group8_sum_first(pot);
        if (l == 0) out[seg] = pot;
    }
}

// after a state import: derive the connected flag of every valid synapse from its permanence
// (the scan reads the flag instead of the permanence row)
__global__ __launch_bounds__(256) void k_tm_flag_connected(Dev d) {
    const int S = d.world > 1 ? d.ctr->L : d.ctr->S;      // (rows)
    const long long total = (long long)S * d.E;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int seg = (int)(i / d.E), slot = (int)(i % d.E);
        if (slot < d.seg_nsyn[seg])
            d.presyn[i] = (d.presyn[i] & SYN_CELL) | (d.sperm[i] >= d.perm_thr ? (int)SYN_CONNECTED : 0);
    }
}

// recount recyclable segments after a state import
__global__ __launch_bounds__(256) void k_tm_recount(Dev d) {
    __shared__ int s_recyc;
    const int S = d.ctr->S, b = blockIdx.x;
    if (b * 1024 >= S) return;
    if (threadIdx.x == 0) s_recyc = 0;
    __syncthreads();
    int v = 0;
    for (int q = 0; q < 4; ++q) {
        int s = b * 1024 + threadIdx.x * 4 + q;
        v += (s < S && d.seg_nsyn[s] < d.match_thr) ? 1 : 0;
    }
    if (v) atomicAdd(&s_recyc, v);
    __syncthreads();
    if (threadIdx.x == 0) {                        // (the second level was zeroed before the launch)
        d.recyc_cnt[b] = s_recyc;
        if (s_recyc) atomicAdd(&d.recyc_cnt2[b >> 10], s_recyc);
    }
}

#endif
