// The fused launches of htm_run's pipelined schedules -- two per timestep (below: k_act_mid_rows, k_learn_scan_emit), three
// (BITHTM_LEAN=1: k_act_rows, k_mid_overlap, k_learn_scan_emit), or four where the scan's column bitmap does not fit the LDS
// -- and the device-clock trace.
// Part of the single translation unit htm_engine.hip (included there, in this order:
// htm_dev.h, htm_sp_kernels.h, htm_tm_kernels.h, htm_pipeline.h).
#ifndef BITHTM_HTM_PIPELINE_H
#define BITHTM_HTM_PIPELINE_H

// ---- pipelined schedules: roles of different steps share every launch --------------------------
// A forked stream / graph branch costs 17-29 us on this runtime and every dependent launch 1.2-3 us
// plus its own chain of memory round trips; heterogeneous blocks in one launch cost nothing.  The
// Spatial Pooler never reads Temporal Memory state, so inside a batched run it works ahead of the
// Temporal Memory.  The four-launch schedule (BITHTM_LEAN=0; t = the TM's step):
//
//   k_open_emit(t)      activation of step t's winner columns     | rest of the select + winner list (t+1)
//   k_mid_rows(t)       segment allocation, learn/punish list     | SP permanence rows + duty cycle (t+1)
//   k_learn_overlap(t)  synapse learning and growth               | overlap + boost + select digit 0 (t+2)
//   k_scan_sel(t)       segment scan                              | select digit 1 (t+2), clears for t+1
//
// The pairing follows what was measured with the device clock (tools/step_timeline.py): the scan's
// gathers fill the memory pipeline and stretch every dependent access of a co-resident wave, and its
// blocks take every CU slot, so it shares its launch only with the lightest SP role; the
// latency-bound select finish runs beside the cheap activation; the two streaming roles (rows,
// overlap) sit beside the latency-bound mid and learn roles.  Here the look-ahead includes the SP's
// persistent updates (rows, duty cycle), so it only happens between two steps of one htm_run
// call: the last two steps of a run look ahead less (StepPlan) and no call returns with SP work
// outstanding.
struct TraceScope {                                 // BITHTM_TRACE=1: first / last device clock of every block
    unsigned long long *t;
    __device__ TraceScope(const Dev &d, int slot) {
#if defined(BITHTM_LEARN_STAMPS) || defined(BITHTM_SCAN_STAMPS) || defined(BITHTM_EMIT_STAMPS) || defined(BITHTM_SHARD_STAMPS)
        t = nullptr;                                // the diagnostic builds of the learning role and of the scan use the buffer
        return;
#endif
        t = (d.trace && blockIdx.x < 4096 && d.ctr->step[slot >> 2] < d.trace_until) ? d.trace + ((size_t)slot * 4096 + blockIdx.x) * 2 : nullptr;
        if (t && threadIdx.x == 0) t[0] = wall_clock64();
    }
    __device__ ~TraceScope() { if (t && threadIdx.x == 0) t[1] = wall_clock64(); }
};

// the emit blocks wait for each other's records: they come first in the grid, so that all of them
// are resident whatever the other blocks do
__global__ __launch_bounds__(256) void k_open_emit(Dev d, int p, int n_emit_blocks, int n_active) {
    TraceScope ts(d, 0 + 4 * p);
    if ((int)blockIdx.x < n_emit_blocks) {
        role_emit(d, p ^ 1, 1, 1, 0, blockIdx.x, n_emit_blocks, (EmitShared *)dyn_lds);
    } else {                                       // one active column per lane group (a half-wave; the wave where a column has 64 cell slots)
        const int idx = ((int)blockIdx.x - n_emit_blocks) * tm_groups_per_block(d) + tm_group_of(d, threadIdx.x);
        const bool ok = idx < n_active;
        const int a = ok ? d.active_cols[p][idx] : 0;
        tm_activate_column(d, p, 1, ok, a, idx, tm_pred_words(d, p, ok, a));
    }
}

// blocks [0, 1 + n_cls): the middle of the TM step; then one SP winner row per block -- of this step
// (rows_ahead = 0: one role per launch) or of the coming one (pipelined schedule) --; then the coming
// step's duty cycle (regularizations.py:19-21, float32, two roundings).  256-thread blocks: the
// dispatcher places them about five times faster, wave for wave, than 1024-thread ones (measured:
// 2000 small blocks start within 1 us, 800 large ones take 7), and all of them are resident at once.
#ifndef BITHTM_MID_ROWS_WAVES
#define BITHTM_MID_ROWS_WAVES 8
#endif
__global__ __launch_bounds__(256, BITHTM_MID_ROWS_WAVES) void k_mid_rows(Dev d, int p, int n_active, int want_winner, int learning, int n_cls,
                                                  const uint32_t *__restrict__ bank, int n_inputs, int n_rows, int rows_ahead, int n_duty_blocks) {
    TraceScope ts(d, 1 + 4 * p);
    int b = blockIdx.x;
    if (b <= n_cls) {
        role_mid<256>(d, p, n_active, want_winner, learning, b, n_cls);
        return;
    }
    b -= 1 + n_cls;
    const int q = p ^ 1;
    if (b < n_rows) {
        role_sp_row<256>(d, rows_ahead ? q : p, bank, n_inputs, rows_ahead, b, threadIdx.x);
        return;
    }
    b -= n_rows;
    const int c = d.c0 + b * 256 + (int)threadIdx.x;              // (own columns)
    if (b < n_duty_blocks) {
        if (c < d.c1) {
            float dc = d.duty[c] * d.mom;
            if ((d.colbits[rows_ahead ? q : p][c >> 5] >> (c & 31)) & 1u) dc = dc + d.dinc;
            d.duty[c] = dc;
        }
        return;
    }
    // the remaining blocks: zero the match bits this step's scan (and learning role) will set with atomicOr -- the buffer
    // holds the bits of the scan two steps back, whose readers are done.  (Rows at or above the segment count never had one.)
    b -= n_duty_blocks;
    const int nz = (int)gridDim.x - 1 - n_cls - n_rows - n_duty_blocks;
    const int words4 = ((d.world > 1 ? d.ctr->L : d.ctr->S) + 127) >> 7;
    uint4 *mb = (uint4 *)d.match_bits[p];
    for (int i = b * 256 + (int)threadIdx.x; i < words4; i += nz * 256) mb[i] = make_uint4(0u, 0u, 0u, 0u);
}

template <int EPL>
__global__ __launch_bounds__(RB) void k_learn_overlap(Dev d, int p, int n_learn_blocks, const uint32_t *__restrict__ bank,
                                                        int n_inputs, int G, int sp, int step_offset) {
    TraceScope ts(d, 2 + 4 * p);
    if ((int)blockIdx.x < n_learn_blocks)
        role_learn<EPL, RB>(d, p, blockIdx.x, n_learn_blocks, (LearnShared<EPL, RB> *)dyn_lds);
    else
        role_overlap<RB>(d, bank, n_inputs, G, p, sp, step_offset, blockIdx.x - n_learn_blocks, gridDim.x - n_learn_blocks, (uint32_t *)dyn_lds);
}

// blocks [0, n_sel): select digit 1 for the SP step with parity sp; then n_clear blocks that zero the
// dense per-column words of the coming step (what EMIT_CLEAR does when the winner list is emitted in a
// launch of its own: here the learning role still needed them after the emit); the rest: the scan.
// (with the learning role and the scan in one launch, k_learn_scan below, this one is left with the SP roles)
// The few short SP blocks come first: behind the scan blocks they would wait for a free CU slot.
template <bool use_lds, int MINW>
__global__ __launch_bounds__(256, MINW) void k_scan_sel(Dev d, int p, int n_sel_blocks, int n_clear_blocks, int sp, int n_spec) {
    TraceScope ts(d, 3 + 4 * p);
    int b = blockIdx.x;
    if (b < n_sel_blocks) {
        role_sel_pass<256>(d, 1, sp, b, n_sel_blocks, (SelShared *)dyn_lds);
        return;
    }
    b -= n_sel_blocks;
    if (b < n_clear_blocks) {
        const int q = p ^ 1;
        for (int c = b * 256 + (int)threadIdx.x; c < d.C * d.WPC; c += n_clear_blocks * 256) {
            d.act[q][c] = 0;
            d.win[q][c] = 0;
            d.pred[q][c] = 0;
        }
        return;
    }
    b -= n_clear_blocks;
    role_scan<256, use_lds, MINW == 1>(d, p, b, gridDim.x - n_sel_blocks - n_clear_blocks, n_spec, (uint32_t *)dyn_lds);
}

// ---- the three-launch schedule -------------------------------------------------------------------------------
// The Temporal Memory's chain is activate -> mid -> learn -> scan, the Spatial Pooler's emit -> rows -> overlap -> select
// digit -> emit; a dependent launch costs about 2 us whatever is in it.  Two changes make both chains three long:
// the learning role and the scan share a launch (role_learn SELF, SEG_BUSY), and the select needs one histogram pass
// (win_bin).  The SP is then one stage ahead of the TM (t = the TM's step):
//
//   k_act_rows(t)      activate(t), dense-word clears of the step   | rows(t) + duty(t)       SP learning of this step
//   k_mid_overlap(t)   mid(t), match-bit zeroing                     | overlap(t+1)            + windowed histogram
//   k_learn_scan_emit(t)  learn(t) + scan(t)                         | emit(t+1)               select finish + winner list
//
// (emit(t+1) only reads: the SP's persistent state -- permanences, duty cycle -- is never ahead of the TM's step.)
__global__ __launch_bounds__(256, 8) void k_act_rows(Dev d, int p, int n_active, int n_act_blocks, const uint32_t *__restrict__ bank, int n_inputs,
                                                     int n_rows, int n_duty_blocks) {
    TraceScope ts(d, 0 + 4 * p);
    int b = blockIdx.x;
    if (b < n_act_blocks) {                         // one active column per lane group (a half-wave; the wave where a column has 64 cell slots)
        const int idx = b * tm_groups_per_block(d) + tm_group_of(d, threadIdx.x);
        const bool ok = idx < n_active;
        const int a = ok ? d.active_cols[p][idx] : 0;
        tm_activate_column(d, p, 1, ok, a, idx, tm_pred_words(d, p, ok, a));
        return;
    }
    b -= n_act_blocks;
    if (b < n_rows) {
        role_sp_row<256>(d, p, bank, n_inputs, 0, b, threadIdx.x);
        return;
    }
    b -= n_rows;
    const int c = b * 256 + (int)threadIdx.x;
    if (b < n_duty_blocks) {
        if (c < d.C) {
            float dc = d.duty[c] * d.mom;
            if ((d.colbits[p][c >> 5] >> (c & 31)) & 1u) dc = dc + d.dinc;
            d.duty[c] = dc;
        }
        return;
    }
    // the step's dense per-column words: predictions (the scan sets bits), and the active / winner words of the columns
    // that are NOT active (the activation blocks of this launch write the others)
    // (word w belongs to column w / WPC)
    const int n_clear = (int)gridDim.x - n_act_blocks - n_rows - n_duty_blocks;
    for (int w = (b - n_duty_blocks) * 256 + (int)threadIdx.x; w < d.C * d.WPC; w += n_clear * 256) {
        const int cc = w >> (d.LK - 5);
        d.pred[p][w] = 0;
        if (!((d.colbits[p][cc >> 5] >> (cc & 31)) & 1u)) { d.act[p][w] = 0; d.win[p][w] = 0; }
    }
}

__global__ __launch_bounds__(256) void k_mid_overlap(Dev d, int p, int n_active, int want_winner, int learning, int n_cls,
                                                     const uint32_t *__restrict__ bank, int n_inputs, int G, int n_overlap_blocks) {
    TraceScope ts(d, 1 + 4 * p);
    int b = blockIdx.x;
    if (b <= n_cls) {
        role_mid<256>(d, p, n_active, want_winner, learning, b, n_cls);
        return;
    }
    b -= 1 + n_cls;
    if (b < n_overlap_blocks) {
        role_overlap<256>(d, bank, n_inputs, G, p, p ^ 1, 1, b, n_overlap_blocks, (uint32_t *)dyn_lds, 1);
        return;
    }
    b -= n_overlap_blocks;
    const int nz = (int)gridDim.x - 1 - n_cls - n_overlap_blocks;      // (see k_mid_rows)
    const int words4 = ((d.world > 1 ? d.ctr->L : d.ctr->S) + 127) >> 7;
    uint4 *mb = (uint4 *)d.match_bits[p];
    for (int i = b * 256 + (int)threadIdx.x; i < words4; i += nz * 256) mb[i] = make_uint4(0u, 0u, 0u, 0u);
}

// ---- the two-launch schedule (the default; BITHTM_LEAN=2) -----------------------------------------------------------
// k_act_rows and k_mid_overlap in ONE launch; the last launch stays k_learn_scan_emit:
//
//   k_act_mid_rows(t)     activate(t) -> fan-in -> mid(t), clears, match-bit zeroing
//                         | rows(t) + each winner column's own overlap(t+1)  | overlap(t+1) of the other columns + duty(t)
//   k_learn_scan_emit(t)  learn(t) + scan(t)                                 | emit(t+1)
//
// Two dependencies used to need the boundary between the two launches.  (1) overlap(t+1) reads the permanence masks rows(t)
// rewrites: only the k winner rows change, and their blocks hold the new connected bits in registers -- they count them against
// the coming input themselves (role_sp_row OWN), the overlap role passes over the winners (role_overlap fold = 2) and takes
// the duty-cycle update along (a column's duty cycle is read and written by the one block that finishes the column).
// (2) mid(t) reads what activate(t) writes: the activation blocks store through to memory, wait for the acknowledgement and
// count themselves done on one of 16 counters; the middle role's blocks -- behind them in the grid, so every activation
// block has been dispatched before any of them can spin -- poll the counters' sum and read the activation's output with
// agent-scope loads (NOT behind an acquire fence: buffer_inv sc1 by every wave of the role took the launch from 11 to 26 us).
// In place the hand-off costs about 2 us (acknowledgement, atomic, poll: three trips to memory under the launch's traffic;
// tools/fanin.hip measured 1.1-1.5 us on an idle device) -- what a launch boundary costs: the gain of the schedule is that the
// winner rows and the overlap now stream UNDER the chain activate -> mid instead of before and after it.
// All of the launch's blocks are resident at once at the bench shape (activation 164 + overlap 512 + middle 33 + rows 1 311 =
// 2 020 of the 2 048 slots 64 registers leave); nothing but the middle role waits for another block, so the order of the grid
// is only a matter of speed.  The counters are reset by k_learn_scan_emit, which always follows.
__device__ __forceinline__ void fan_signal(const Dev &d, int p, int b) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (the write-through stores are acknowledged)
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_fetch_add(d.fan + (size_t)(p * FAN_COUNTERS + (b & (FAN_COUNTERS - 1))) * FAN_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void fan_wait(const Dev &d, int p, uint32_t target) {
    if (threadIdx.x < 64) {
        for (int looks = 0;; ++looks) {
            uint32_t v = threadIdx.x < FAN_COUNTERS
                             ? __hip_atomic_load(d.fan + (size_t)(p * FAN_COUNTERS + (int)threadIdx.x) * FAN_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            v = wave_sum(v);
            if (v >= target) break;
            if (looks >= (1 << 18)) {                                // (every wave leaves: an activation block never arrived)
                if (threadIdx.x == 0) atomicOr(&d.ctr->error, 32);
                break;
            }
#ifndef BITHTM_FAN_SLEEP
#define BITHTM_FAN_SLEEP 4
#endif
            __builtin_amdgcn_s_sleep(BITHTM_FAN_SLEEP);
        }
    }
    __syncthreads();                                                // (what the activation wrote is read with agent-scope loads: role_mid, same_launch)
}

// grid: [activation][middle role: block 0, classification][winner rows][overlap | duty cycle][dense-word clears][match-bit zeroing]
// n_overlap_blocks > 0: the coming step's overlap rides along (a steady-state step); else the duty cycle has blocks of its own
// and the rows are plain (the last step of a call)
#ifndef BITHTM_LEAN2_WAVES
#define BITHTM_LEAN2_WAVES 8
#endif
__global__ __launch_bounds__(256, BITHTM_LEAN2_WAVES) void k_act_mid_rows(Dev d, int p, int n_active, int n_act_blocks, int learning, int n_cls,
                                                      const uint32_t *__restrict__ bank, int n_inputs, int n_rows, int G, int n_overlap_blocks,
                                                      int n_duty_blocks, int n_clear_blocks, int order) {
    TraceScope ts(d, 0 + 4 * p);
    int b = blockIdx.x;
    // the four big roles in the order the host asks for (hex digits, first role first: 0 activation, 1 middle, 2 rows, 3 overlap;
    // the activation before the middle role, which waits for it)
    int role = 4;
#pragma unroll
    for (int i = 3; i >= 0 && role == 4; --i) {
        const int r = (order >> (4 * i)) & 15;
        const int n = r == 0 ? n_act_blocks : r == 1 ? 1 + n_cls : r == 2 ? n_rows : n_overlap_blocks;
        if (b < n) role = r; else b -= n;
    }
    if (role == 0) {
        const int idx = b * tm_groups_per_block(d) + tm_group_of(d, threadIdx.x);
        const bool ok = idx < n_active;
        const int a = ok ? d.active_cols[p][idx] : 0;
        tm_activate_column<true>(d, p, 1, ok, a, idx, tm_pred_words(d, p, ok, a));
        fan_signal(d, p, b);
        return;
    }
    if (role == 1) {
        role_mid<256>(d, p, n_active, 1, learning, b, n_cls, 1, [&]() { fan_wait(d, p, (uint32_t)n_act_blocks); });
        return;
    }
    if (role == 2) {
        if (n_overlap_blocks > 0) role_sp_row<256, true>(d, p, bank, n_inputs, 0, b, threadIdx.x);
        else role_sp_row<256>(d, p, bank, n_inputs, 0, b, threadIdx.x);
        return;
    }
    if (role == 3) {
        role_overlap<256>(d, bank, n_inputs, G, p, p ^ 1, 1, b, n_overlap_blocks, (uint32_t *)dyn_lds, 1, n_rows > 0 ? 2 : 1);
        return;
    }
    if (b < n_duty_blocks) {
        const int c = b * 256 + (int)threadIdx.x;
        if (c < d.C) {
            float dc = d.duty[c] * d.mom;
            if ((d.colbits[p][c >> 5] >> (c & 31)) & 1u) dc = dc + d.dinc;
            d.duty[c] = dc;
        }
        return;
    }
    b -= n_duty_blocks;
    if (b < n_clear_blocks) {                          // (see k_act_rows)
        for (int w = b * 256 + (int)threadIdx.x; w < d.C * d.WPC; w += n_clear_blocks * 256) {
            const int cc = w >> (d.LK - 5);
            d.pred[p][w] = 0;
            if (!((d.colbits[p][cc >> 5] >> (cc & 31)) & 1u)) { d.act[p][w] = 0; d.win[p][w] = 0; }
        }
        return;
    }
    b -= n_clear_blocks;
    const int nz = (int)gridDim.x - n_act_blocks - 1 - n_cls - n_rows - n_overlap_blocks - n_duty_blocks - n_clear_blocks;      // (see k_mid_rows)
    const int words4 = (d.ctr->S + 127) >> 7;
    uint4 *mb = (uint4 *)d.match_bits[p];
    for (int i = b * 256 + (int)threadIdx.x; i < words4; i += nz * 256) mb[i] = make_uint4(0u, 0u, 0u, 0u);
}

// the emit blocks wait for each other's records: they come first in the grid (all resident whatever the others do);
// then the learning role, whose items are the longest chains; then the scan
// TAB: the scan looks active cells up in the LDS tables of the step's select finish and activation (role_scan) -- the
// three-launch schedule, whose every step has them; the launch without emit blocks (enqueue_tm: stand-alone Temporal
// Memory, shards) reads the cell words from memory
// MINW < 6: the large-pool form: the scan streams, the grid is what is resident at once, and a block whose select finish or
// learning items are done JOINS the scan (role_scan, DYN) instead of leaving its slot to a scan block that would have to be
// dispatched and stage the bitmap first.  n_scan_blocks < 0: scan blocks with fixed shares and nothing joining, as before
// (BITHTM_SCAN_DYN=0).
template <int EPL, int MINW, bool TAB = false>
__global__ __launch_bounds__(256, MINW) void k_learn_scan_emit(Dev d, int p, int n_emit_blocks, int n_learn_blocks, int n_scan_blocks, int n_spec) {
    TraceScope ts(d, 2 + 4 * p);
    constexpr bool LARGE = MINW < 6;
    static_assert(!(LARGE && TAB), "the streaming scan reads the cell words from memory");
    const bool dyn = LARGE && n_scan_blocks > 0;
    if (n_scan_blocks < 0) n_scan_blocks = -n_scan_blocks;
    // (the fan-in counters of the step's first launch, k_act_mid_rows, are this launch's to reset: it always follows that one, whatever
    // schedule the steps before and after take)
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x < FAN_COUNTERS) d.fan[(size_t)(p * FAN_COUNTERS + (int)threadIdx.x) * FAN_STRIDE] = 0u;
    int b = blockIdx.x, scan_blk;
    if (b < n_emit_blocks) {
        role_emit(d, p ^ 1, 1, 1, 0, b, n_emit_blocks, (EmitShared *)dyn_lds, 1);
        if (!dyn) return;
        scan_blk = -1 - b;                           // (joins the scan)
    } else if (b < n_emit_blocks + n_learn_blocks) {
        role_learn<EPL, 256, true>(d, p, b - n_emit_blocks, n_learn_blocks, (LearnShared<EPL, 256> *)dyn_lds);
        if (!dyn) return;
        scan_blk = -1 - b;
    } else {
        scan_blk = b - n_emit_blocks - n_learn_blocks;
    }
    // (one call site: the role is the launch's largest piece of code, and four inlined copies of it spilled)
    if (LARGE) role_scan<256, true, true, false, true>(d, p, scan_blk, n_scan_blocks, dyn ? n_emit_blocks + n_learn_blocks : 0, (uint32_t *)dyn_lds);
    else role_scan<256, true, false, TAB>(d, p, scan_blk, n_scan_blocks, n_spec, (uint32_t *)dyn_lds);
}

// The last launch of a step run role by role (enqueue_tm): the learning role, the scan, and a streaming role behind them --
//   n_rows > 0   the Spatial Pooler's permanence rows of THIS step (a host-fed step, htm_step: they used to share the middle
//                launch, whose block 0 is the step's longest chain; beside the learning role and the scan they are free)
//   else         the overlap of the COMING step on a shard's own columns (htm_shard_run; their permanence rows and duty
//                cycle were updated by this step's middle launch; nothing of the Temporal Memory is read)
template <int EPL>
__global__ __launch_bounds__(256, 6) void k_learn_scan_tail(Dev d, int p, int n_learn_blocks, int n_scan_blocks, int n_spec,
                                                            const uint32_t *__restrict__ bank, int n_inputs, int G, int wmode, int n_rows) {
    int b = blockIdx.x;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x < FAN_COUNTERS) d.fan[(size_t)(p * FAN_COUNTERS + (int)threadIdx.x) * FAN_STRIDE] = 0u;     // (as k_learn_scan_emit)
    if (b < n_learn_blocks) {
        role_learn<EPL, 256, true>(d, p, b, n_learn_blocks, (LearnShared<EPL, 256> *)dyn_lds);
        return;
    }
    b -= n_learn_blocks;
    if (b < n_scan_blocks) {
        role_scan<256, true, false, false>(d, p, b, n_scan_blocks, n_spec, (uint32_t *)dyn_lds);
        return;
    }
    b -= n_scan_blocks;
    if (n_rows > 0) role_sp_row<256>(d, p, bank, n_inputs, 0, b, threadIdx.x);
    else role_overlap<256>(d, bank, n_inputs, G, p, p ^ 1, 1, b, (int)gridDim.x - n_learn_blocks - n_scan_blocks, (uint32_t *)dyn_lds, wmode);
}

// A host-fed step's first launch when the PREVIOUS step's last one was held back (htm_step, one input per call): that
// step's learning role and scan beside this step's overlap.  The overlap needs the previous step's permanence rows -- they
// rode in its middle launch -- and nothing of the Temporal Memory; the input comes with the launch's arguments.
template <int EPL>
__global__ __launch_bounds__(256, 6) void k_learn_scan_front(Dev d, int p_prev, int n_learn_blocks, int n_scan_blocks, int n_spec,
                                                             PackedInputArg in, int G, int p, int wmode) {
    int b = blockIdx.x;
    // (the fan-in counters of the held-back step's k_act_mid_rows: see k_learn_scan_emit)
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x < FAN_COUNTERS) d.fan[(size_t)(p_prev * FAN_COUNTERS + (int)threadIdx.x) * FAN_STRIDE] = 0u;
    if (b < n_learn_blocks) {
        role_learn<EPL, 256, true>(d, p_prev, b, n_learn_blocks, (LearnShared<EPL, 256> *)dyn_lds);
        return;
    }
    b -= n_learn_blocks;
    if (b < n_scan_blocks) {
        role_scan<256, true, false, false>(d, p_prev, b, n_scan_blocks, n_spec, (uint32_t *)dyn_lds);
        return;
    }
    b -= n_scan_blocks;
    uint32_t *h = (uint32_t *)dyn_lds, *s_in = h + SEL_BINS;       // (SEL_BINS * 4 is a multiple of 16: the row loads are 16 bytes)
    if (threadIdx.x < ARG_INPUT_WORDS) {
        const uint32_t v = (int)threadIdx.x < d.W ? in.w[threadIdx.x] : 0u;
        s_in[threadIdx.x] = v;
        if (b == 0 && (int)threadIdx.x < d.W) d.input_stage[threadIdx.x] = v;
    }
    __syncthreads();
    role_overlap<256>(d, s_in, 1, G, p, p, 0, b, (int)gridDim.x - n_learn_blocks - n_scan_blocks, h, wmode);
}

#endif
