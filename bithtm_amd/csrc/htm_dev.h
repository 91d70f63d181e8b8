// Constants, the device counter block, the kernel argument struct and small device helpers.
// Part of the single translation unit htm_engine.hip (included there, in this order:
// htm_dev.h, htm_sp_kernels.h, htm_tm_kernels.h, htm_pipeline.h).
#ifndef BITHTM_HTM_DEV_H
#define BITHTM_HTM_DEV_H

typedef unsigned long long u64;

#define SEL_MAX_PASSES 6
#define SEL_DIGIT 12
#define SEL_BINS 4096         // 1 << SEL_DIGIT
#define HIST_REP 4            // copies of the digit-0 histogram (its few hot bins take one atomic per block; 8 and 2 measured no better)
#define SEL_COARSE 64         // ... and the sums of the 64 runs of 64 bins, in COARSE_REP copies of their own (every block of the overlap adds
#define COARSE_REP 16         // to a few hot runs: block b to copy b % COARSE_REP): the windowed select picks the run first, then the bin
                              // inside it -- two reads of a few kilobytes instead of one of 64 KB by every block of the select's finish
#define COARSE_STRIDE 1024    // words between two coarse copies (a page each: the copies' atomics go to different memory channels)
#define FAN_COUNTERS 16       // the in-launch fan-in of the two-launch schedule (k_act_mid_rows): producer block b adds to counter b % 16,
#define FAN_STRIDE 32         // the counters 128 bytes apart (tools/fanin.hip: 1.1-1.5 us from the last producer's bytes to the consumers)
#define HIST0_FINE (HIST_REP * SEL_BINS)
#define HIST0_PAR (HIST0_FINE + COARSE_REP * COARSE_STRIDE)      // words per step parity: fine copies, then coarse copies
#define RB 512                // threads per block of the role kernels (overlap, select, learn, scan)
#define SCAN_SEGS 64          // segments per 256-thread block iteration of the segment scan
#define DEAD_CAP 256          // newly dead segment ids one rank can report per exchange
#define SHARD_HOT_KEYS 4      // hot-list keys per thread of the sharded step's global select (1024 threads: 4 096 between the ranks)
#define CAND_CAP 256          // growth candidates staged per wave
#define WIN_LDS 4096          // previous winner cells the learning role keeps in LDS (more are read from global memory)
#define MAX_SLOTS 512
#define SYN_CONNECTED 0x80000000u   // a presynaptic id carries `permanence >= threshold` in its top bit ...
#define SYN_CELL 0x7FFFFFFF         // ... and the cell (column * 32 + cell) below it
#define SEG_BUSY 0x40000000         // in seg_nsyn: the segment is on this step's work list -- the learning role is about to
                                    // rewrite its row (and clears the flag with the new count); the scan leaves it alone
#define EPS32 1e-8f           // `epsilon=1e-8` against float32 arrays (weak Python scalar): the default of Dev::eps

// Select key of a boosted overlap: an order-preserving image of the double's bits that spends 8 bits on the
// exponent instead of 12 on sign + exponent.  boosted = (float32 factor) x (overlap <= input_dim <= 2^17) is 0 or
// lies in [2^-149, 2^17], 167 exponents, so the exponent field minus (1023 - 150) fits 8 bits and the whole key
// moves up by 3: two launched 12-bit digits then resolve 3 more mantissa bits, and the bucket the select
// finishes in-kernel is 8 times narrower (a crowded bucket is what makes `emit` slow).
#define KEY_SHIFT 3
__device__ __forceinline__ u64 select_key_bits(u64 bits) { return bits ? (bits - ((u64)(1023 - 150) << 52)) << KEY_SHIFT : 0ull; }
__device__ __forceinline__ u64 select_key(double boosted) { return select_key_bits((u64)__double_as_longlong(boosted)); }

// The windowed select of the three-launch schedule: ONE histogram pass.  The bins are not the key's top 12 bits but a
// monotone function of the key that spends them where the k-th key is expected -- WIN_COARSE values of the top digit
// starting at `base` (the previous step's k-th key minus half the window), each split 128 ways by the next 7 bits
// (19 key bits resolved inside the window), one bin for everything below and one for everything above.  The k-th
// boosted overlap of a learned pattern is up to 1.5 times that of a new one (measured: a window of a factor 1.44 either
// way missed 7 % of the bench workload's steps); this window spans a factor of 2.6 either way.  If the key falls outside,
// the select finishes by the exact fallback of role_emit (slow, rare, same result).
#define WIN_COARSE 31
#define WIN_FINE 7
#define WIN_BINS (2 + (WIN_COARSE << WIN_FINE))
#define WIN_LOWBITS (52 - WIN_FINE)        // key bits left unresolved inside the window
static_assert(WIN_BINS <= SEL_BINS, "the window's bins fit the select histogram");
__host__ __device__ __forceinline__ uint32_t win_bin(u64 key, uint32_t base) {
    const uint32_t c = (uint32_t)(key >> 52);
    if (c < base) return 0u;
    const uint32_t o = c - base;
    return o >= WIN_COARSE ? 1u + (WIN_COARSE << WIN_FINE) : 1u + ((o << WIN_FINE) | ((uint32_t)(key >> WIN_LOWBITS) & ((1u << WIN_FINE) - 1u)));
}
__host__ __device__ __forceinline__ uint32_t win_base_for(u64 kth_key) {
    const uint32_t c = (uint32_t)(kth_key >> 52);
    return c > WIN_COARSE / 2 ? min(c - WIN_COARSE / 2, 4096u - WIN_COARSE) : 0u;
}

// The window of the column-sharded step's GLOBAL select (k_shard_select), around the previous step's global k-th key: the
// same shape as the local one.  Keys below it are not counted (the pick counts from the top; if it gets that far, the
// digit passes take over).
__host__ __device__ __forceinline__ uint32_t win_base_global(u64 kth_key) { return win_base_for(kth_key); }

// radix-select digit p covers key bits [shift, shift + bits): 12 bits from the top, the last one 4
__host__ __device__ __forceinline__ int sel_shift(int pass) { return pass < 5 ? 52 - SEL_DIGIT * pass : 0; }
__host__ __device__ __forceinline__ int sel_bits(int pass) { return pass < 5 ? SEL_DIGIT : 4; }

// ------------------------------------------------------------------------------------------
// device-resident scalars
struct Counters {
    uint32_t step[2];         // timestep index (key of the random draws); step t reads step[t & 1]
                              // and its last kernel writes step[(t + 1) & 1] = t + 1
    int32_t S;                // allocated segment ids
    int32_t n_win[2];         // winner cells of step parity p
    int32_t has_winner[2];    // winner list of parity p is valid (winner_cell is not None)
    int32_t has_distal;       // a scan has run (distal_state is not None)
    uint32_t cm_dense_step;   // 1 + the step whose learning role clears ALL of the previous per-cell maxima, not just those of
                              // the matching segments: set by a state import (the imported maxima need not belong to the
                              // cells the imported matching segments belong to now)
    int32_t n_active_cells;
    // (by step parity: the scan of step t, which resets the counts for step t + 1, may share its launch with the
    // learning role of step t, which reads them)
    alignas(128) int32_t n_work[2];   // learning / punish work items of the step (front of the work array): the one field of this
                              // block that takes atomics from many blocks -- on a line of its own, away from the scalars everybody reads
    int32_t n_bind[2];        // newly bound segments of the step (back of the work array, growing down)
    alignas(128) int32_t n_work_last;      // ... of the last completed step (telemetry)
    int32_t sel_fallbacks;    // steps whose top-k select took the in-kernel fallback (telemetry)
    int32_t cand_exact;       // sharded: steps whose LOCAL select cut the threshold bin exactly (record exchange) instead of
                              // handing the whole bin over (telemetry)
    int32_t hot_selects;      // sharded: steps whose GLOBAL select was settled among the ranks' hot lists (telemetry)
    int32_t sel_zooms;        // steps whose select finish cut a crowded threshold bin to the k-th key's sub-bin (telemetry)
    uint32_t emit_epoch;      // bumped by every overlap launch: tags the records k_sp_emit's blocks exchange
    int32_t n_un;             // winners needing a new segment
    int32_t n_recycled, n_new, S_old;
    int32_t L;                // column-sharded handles: local rows in use are [0, L) (unsharded: local row = segment id, L = S)
    int32_t n_lfree;          // ... and the stack of free local rows below L (rows whose segment was recycled by another rank)
    int32_t error;            // sticky capacity flags
    // Spatial Pooler select state, double-buffered by the parity of the step it belongs to (the
    // pipelined schedule computes step t+1's overlap / select digits while step t's TM runs)
    u64 sel_prefix[2];        // k-th largest key and how many of the keys equal to it are winners
    uint32_t sel_krem[2];
    uint32_t sel_win[2];      // windowed select (win_bin): lowest top-digit value of the window for the SP step of parity p,
                              // set from the k-th key of the step before
    uint32_t sel_win_global;  // column-sharded handles: the same for the global select over the ranks' candidates (k_shard_select)
    u64 sel_pass_prefix[2][SEL_MAX_PASSES + 1];    // radix-select state entering pass p
    uint32_t sel_pass_krem[2][SEL_MAX_PASSES + 1];
};

struct Dev {
    int I, W, W4, Ipad, C, K, k, E, Scap, work_cap, sel_passes, colwords, low_zero, cand_d, cand_pairwise, cand_others, cand_speculate, cand_take_all, cand_zoom, poll_delay;
    int win_offset;           // test knob: added to the select window's base (a window that misses: the fallback every step)
    int cls_rows_max;         // test knob: pools of more rows than this classify by 32-row words (role_mid); -1 = the default, 8 rows per thread of the launch
    int LK, KP, WPC;          // cells per column, padded: KP = 32 (cell_dim <= 32) or 64 cell slots, LK = log2(KP), WPC = KP / 32 words per
                              // column.  A cell is enc = column * KP + cell; the dense cell words (act, pred, win) are indexed by
                              // enc >> 5 = column * WPC + (cell >> 5), the per-cell arrays by enc
    int world, c0, c1;        // this rank owns columns [c0, c1) (world == 1: everything)
    int sel_lo, sel_hi, sel_k; // the select works on the keys of columns [sel_lo, sel_hi) and finds their sel_k largest
                              // (unsharded: all columns, k; a shard selects its own candidates: [c0, c1), min(k, c1 - c0))
    int Lcap, n_cand;         // sharded: local row capacity; candidates a rank must offer = sel_k = min(k, own columns)
    int hot_budget, hot_target;   // ... entries of a rank's hot list the global select looks at; the list goes down to the bin of the
                              // rank's hot_target-th largest key
    int cand_cap;             // ... and the candidate slots of its exchange record (>= n_cand: the local select may hand over the
                              // whole threshold bin instead of cutting it, see role_emit)
    double sp_thr, sp_don, sp_doff;
    float coef, mom, dinc;
    double lrn_act, lrn_inact, pun_act, pun_inact;
    int lrn_prune, pun_prune;
    float perm_init, perm_thr;
    float eps;                // TemporalMemory.process(epsilon=...) (networks.py:91), as float32: 1e-8 unless htm_set_epsilon says otherwise
    int act_thr, match_thr, sample;
    uint32_t seed;
    // Spatial Pooler
    double *perm;             // [C][Ipad] float64 permanences (projections.py:16)
    uint32_t *mask;           // [C][W]    bit-packed `permanence >= threshold` (projections.py:19)
    float *duty;              // [C]
    int *overlap[2];          // [C]   parity double buffer, like the select state
    double *boosted[2];       // [C]
    u64 *key[2];              // [C] bits of boosted (non-negative doubles order like uint64)
    uint32_t *hist;           // [2][SEL_MAX_PASSES][SEL_BINS]  (digits 1..)
    uint32_t *hist0;          // [2][HIST_REP][SEL_BINS + SEL_COARSE]   digit 0 (or the window's bins): block b adds to copy b % HIST_REP
    uint32_t *sel_blk;        // [ceil(C/256)] packed (greater, equal) counts per 256-column block
    uint32_t *sel_rec;        // [ceil(C/256)][32] per-block bucket records of k_sp_emit (16 granules)
    int *active_cols[2];      // [k] ascending; parity double buffer (the pipelined schedule emits step t+1's
                              // list while step t's scan still reads its own)
    uint32_t *input_stage;    // [W] host-fed input
    // Temporal Memory
    uint32_t *act[2];         // [C * WPC] active-cell words, parity double buffer
    uint32_t *pred[2];        // [C * WPC] predicted-cell words
    uint32_t *win[2];         // [C * WPC] winner-cell words
    uint32_t *colbits[2];     // [ceil(C/64)*2] bitmap of the step's active columns
    int *winners[2];          // [k * KP] winner cells (enc), ascending
    uint8_t *bursting;        // [k]
    // per word slot s = idx * WPC + h of the step's active columns (idx-th of the ascending list, its h-th word):
    int *actw_id;             // [k * WPC] the word's index in the dense arrays: active_cols[idx] * WPC + h
    uint32_t *winw_idx;       // [k * WPC] winner word (same as win[actw_id[s]])
    uint8_t *actcnt;          // [k * WPC] popc(active word)
    uint32_t *act_list;       // [k * WPC] active word (same as act[p][actw_id[s]])
    uint16_t *col_rank[2];    // [colwords] active columns below each 32-column word of colbits[p] (written by the select's finish)
    uint32_t *unacc_word;     // [k * WPC]
    uint32_t *fan;            // [2][FAN_COUNTERS * FAN_STRIDE] activation blocks done, per step parity (two-launch schedule: the middle role waits for them)
    int *unacc_list;          // [k * KP] winners without a matching segment, ascending
    int *seg_cell;            // [Scap] owning cell (enc)
    int *seg_nsyn;            // [Scap] valid synapses; rows are packed: slots [0, nsyn) are valid
    int *presyn;              // [Scap][E] presynaptic cell (enc)
    float *sperm;             // [Scap][E] float32 permanence
    int *segcount;            // [C * KP] segments per cell
    // results of the scan of step t live in buffer t & 1: step t + 1 reads them (activation, classification) while its
    // own scan, which may share a launch with its learning role, accumulates into the other one
    uint32_t *cellmax[2];     // [C * KP] float bits of max jittered potential per cell (0 = none); entries are cleared by the
                              // learning role of the step after the scan that set them
    uint32_t *match_bits[2];  // [ceil(Scap/32)] segment is matching (projections.py:247); zeroed by the middle launch of the
                              // step whose scan (and learning role) then set bits with atomicOr
    uint32_t *seg_info;       // [Scap] last scan, MATCHING segments only: potential | activation << 12 | 1 << 30 | active << 31
    float *seg_jit;           // [Scap] jittered potential of the matching segments
    uint32_t *work;           // [work_cap] segment | mode << 31 (0 = learn + grow, 1 = punish)
    int *recyc_cnt;           // [ceil(Scap/1024)] recyclable segments per 1024-segment block
    int *recyc_cnt2;          // [ceil(Scap/2^20)] ... and per 1024 of those blocks: the allocation looks at these first (a pool of
                              // 134 M segments has 130 k block counts, nearly all of them zero)
    int *recyc_need;          // [2*k*32] (block, first rank) pairs of the blocks add_output draws from
    // Column sharding (world > 1).  Segment ids are global (they key the random draws and order the recycling, and
    // must be what an unsharded run would assign), rows are local: a rank stores only the segments of its own
    // cells, in local rows, and keeps for the whole id space only one bit per id (fewer synapses than the matching
    // threshold: what the recycling rule of projections.py:80-81 looks at) and the row of the ids it owns.
    int *seg_gid;             // [Lcap]  global id of a local row, -1 = free (null on unsharded handles: row == id)
    int *g2l;                 // [Scap]  local row of an owned id, -1 otherwise
    uint32_t *dead_bits;      // [Scap/32] replicated on every rank, changed only by identical decisions
    int *lfree;               // [Lcap]  stack of free local rows
    int *asg_gid;             // [k*32]  ids assigned to this step's new segment requests, in request order
    int *cand_cols;           // [sel_k] own candidate columns of this step, ascending
    int *dead_list;           // [1 + DEAD_CAP]: count, global ids of own segments that died while learning
    const uint32_t *punish;   // htm_tm_update: per-cell punishment mask, one word per column (projections.py:269's output_punishment);
                              // null: the cells of columns that are not active (networks.py:107-108,111)
    unsigned char *send;      // this step's exchange record (set per call)
    Counters *ctr;
    unsigned long long *trace;    // [8][4096][2] BITHTM_TRACE=1: device clock at the start / end of every block of the
                                  // pipelined launches (slot = launch + 4 * step parity), else null
    uint32_t trace_until;         // ... of steps with an index below this (BITHTM_TRACE_UNTIL; default: all)
};

// ------------------------------------------------------------------------------------------
// small device helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// Inclusive prefix sum over the 64 lanes with DPP row shifts and broadcasts (seven dependent VALU ops;
// the __shfl_up loop compiles to six ds_bpermute round trips through the LDS crossbar).
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    uint32_t t = v;
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, false);   // row_shr:3
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x114, 0xf, 0xe, false);   // row_shr:4 bank_mask:0xe
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x118, 0xf, 0xc, false);   // row_shr:8 bank_mask:0xc
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x142, 0xa, 0xf, false);   // row_bcast:15 row_mask:0xa
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x143, 0xc, 0xf, false);   // row_bcast:31 row_mask:0xc
    return t;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v), 63); }

// The value lane `src` holds, src the same in every lane (a ballot's first set bit, a constant): v_readlane -- __shfl
// compiles to ds_bpermute, a round trip through the LDS crossbar, whatever the source lane is.
#ifdef BITHTM_BISECT_A
__device__ __forceinline__ uint32_t wave_read(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src); }
__device__ __forceinline__ int wave_read(int v, int src) { return __shfl(v, src); }
#else
__device__ __forceinline__ uint32_t wave_read(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ int wave_read(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
#endif
__device__ __forceinline__ u64 wave_read(u64 v, int src) {
#ifdef BITHTM_BISECT_A
    return ((u64)(uint32_t)__shfl((int)(v >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)v, src);
#endif
    return ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
}

// op over the 64 lanes' 64-bit values (op associative and commutative, ident its identity), the result in every lane: the
// DPP pattern of wave_incl_scan, two moves and one op per step, instead of six pairs of ds_bpermute.
template <typename Op>
__device__ __forceinline__ u64 wave_reduce64(u64 v, u64 ident, Op op) {
    const int il = (int)(uint32_t)ident, ih = (int)(ident >> 32);
#define BITHTM_DPP64(src, ctrl, rm, bm) \
    (((u64)(uint32_t)__builtin_amdgcn_update_dpp(ih, (int)((src) >> 32), ctrl, rm, bm, false) << 32) | \
     (uint32_t)__builtin_amdgcn_update_dpp(il, (int)(uint32_t)(src), ctrl, rm, bm, false))
    u64 t = v;
    t = op(t, BITHTM_DPP64(v, 0x111, 0xf, 0xf));
    t = op(t, BITHTM_DPP64(v, 0x112, 0xf, 0xf));
    t = op(t, BITHTM_DPP64(v, 0x113, 0xf, 0xf));
    t = op(t, BITHTM_DPP64(t, 0x114, 0xf, 0xe));
    t = op(t, BITHTM_DPP64(t, 0x118, 0xf, 0xc));
    t = op(t, BITHTM_DPP64(t, 0x142, 0xa, 0xf));
    t = op(t, BITHTM_DPP64(t, 0x143, 0xc, 0xf));
#undef BITHTM_DPP64
    return wave_read(t, 63);
}

// Sum over each aligned group of 8 lanes, valid in the group's FIRST lane only (row_shl:1,2,3 then 4:
// four VALU ops instead of three ds_bpermute round trips).
__device__ __forceinline__ int group8_sum_first(int v) {
    int t = v;
    t += __builtin_amdgcn_update_dpp(0, v, 0x101, 0xf, 0xf, false);
    t += __builtin_amdgcn_update_dpp(0, v, 0x102, 0xf, 0xf, false);
    t += __builtin_amdgcn_update_dpp(0, v, 0x103, 0xf, 0xf, false);
    t += __builtin_amdgcn_update_dpp(0, t, 0x104, 0xf, 0xf, false);
    return t;
}

// Sum over each aligned group of 8 lanes, in every lane of the group (quad_perm xor 1, xor 2, then row_half_mirror:
// lane i of a row's half reads lane 7 - i, which is in the other quad)
__device__ __forceinline__ int group8_sum_all(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);
    return v;
}

// A 4-byte store that goes through to memory (sc0 sc1), for data another XCD's waves may read within the same launch: a
// plain store stays in the writer's L2 until the line is evicted or the kernel ends.
__device__ __forceinline__ void store_through(int *p, int v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_through(float *p, float v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }

// A block barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope release + barrier + acquire: the
// release waits for EVERY outstanding vector-memory operation of the wave -- gfx9's vmcnt counts loads, stores and atomics
// alike --, so a barrier behind a global store (a published record, a cleared histogram, a winner's words) or behind
// loads that are not needed yet costs their round trip, about a microsecond.  Where the threads of a block only talk
// through LDS, this is the barrier: LDS (and scalar) operations drained, then s_barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// op over each half of the wave (32 lanes: the cells of one column), floats, the result in every lane of the half: the row
// scan of wave_incl_scan, row 0's (2's) result broadcast into row 1 (3), then lanes 31 and 63 read out -- eight VALU
// operations instead of five ds_bpermute round trips.
template <typename Op>
__device__ __forceinline__ float half_reduce(float v, float ident, Op op) {
    const int id = __float_as_int(ident);
#define BITHTM_DPPF(src, ctrl, rm, bm) __int_as_float(__builtin_amdgcn_update_dpp(id, __float_as_int(src), ctrl, rm, bm, false))
    float t = v;
    t = op(t, BITHTM_DPPF(v, 0x111, 0xf, 0xf));
    t = op(t, BITHTM_DPPF(v, 0x112, 0xf, 0xf));
    t = op(t, BITHTM_DPPF(v, 0x113, 0xf, 0xf));
    t = op(t, BITHTM_DPPF(t, 0x114, 0xf, 0xe));
    t = op(t, BITHTM_DPPF(t, 0x118, 0xf, 0xc));
    t = op(t, BITHTM_DPPF(t, 0x142, 0xa, 0xf));
#undef BITHTM_DPPF
    const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 31));
    const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 63));
    return lane_id() < 32 ? lo : hi;
}

template <int BS, bool LDS_ONLY = false>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_wave, uint32_t &total) {
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t x = wave_incl_scan(v);
    if (lane == 63) s_wave[wv] = x;
    if (LDS_ONLY) lds_barrier(); else __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < BS / 64; ++i) {
        uint32_t t = s_wave[i];
        if (i < wv) woff += t;
        tot += t;
    }
    if (LDS_ONLY) lds_barrier(); else __syncthreads();
    total = tot;
    return woff + x - v;
}

// all lanes of the wave must call; returns the slot for lanes with pred, -1 otherwise
__device__ __forceinline__ int wave_append(int *counter, bool pred) {
    u64 m = __ballot(pred);
    if (m == 0) return -1;
    int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane_id() == leader) base = atomicAdd(counter, __popcll(m));
    base = wave_read(base, leader);
    return pred ? base + __popcll(m & lanemask_lt()) : -1;
}

// recyclable-segment counts, both levels (blk1024 = id >> 10)
__device__ __forceinline__ void recyc_add(const Dev &d, int blk1024, int delta) {
    atomicAdd(&d.recyc_cnt[blk1024], delta);
    atomicAdd(&d.recyc_cnt2[blk1024 >> 10], delta);
}

// h[digit] += 1 for every lane with `active`, one LDS atomic per distinct digit in the wave (keys
// of neighbouring columns mostly share their leading digits: per-lane atomics would serialise).
// All lanes of the wave must call.
__device__ __forceinline__ void hist_add(uint32_t *h, uint32_t digit, bool active) {
    u64 todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t dl = wave_read(digit, leader);
        const u64 same = __ballot(active && digit == dl) & todo;
        if (lane_id() == leader) atomicAdd(&h[dl], (uint32_t)__popcll(same));
        todo &= ~same;
    }
}

// The same for keys that are mostly different but may hold one value many times (a many-way tie): the copies of the
// first active lane's digit are counted with one atomic (same-address LDS atomics run one after the other), the other
// lanes add theirs as they are.  All lanes of the wave must call.
__device__ __forceinline__ void hist_add_tie(uint32_t *h, uint32_t digit, bool active) {
    const u64 act = __ballot(active);
    if (!act) return;
    const int leader = __ffsll((long long)act) - 1;
    const uint32_t dl = wave_read(digit, leader);
    const u64 same = __ballot(active && digit == dl);
    if (lane_id() == leader) atomicAdd(&h[dl], (uint32_t)__popcll(same));
    if (active && digit != dl) atomicAdd(&h[digit], 1u);
}

// bits of v moved to the even bit positions of a 64-bit word
__device__ __forceinline__ u64 spread32(uint32_t v) {
    u64 x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}

__device__ __forceinline__ uint32_t cell_mask(int K) { return K >= 32 ? 0xFFFFFFFFu : ((1u << K) - 1u); }
__device__ __forceinline__ u64 cell_mask64(int K) { return K >= 64 ? ~0ull : ((1ull << K) - 1ull); }
// the valid cells of the h-th word of a column
__device__ __forceinline__ uint32_t cell_mask_word(int K, int h) { return cell_mask(K - 32 * h); }
// the reference's flat cell id (column * cell_dim + cell) of an encoded cell (column * KP + cell)
__device__ __forceinline__ uint32_t enc_to_flat(const Dev &d, int enc) { return (uint32_t)((enc >> d.LK) * d.K + (enc & (d.KP - 1))); }

#endif
