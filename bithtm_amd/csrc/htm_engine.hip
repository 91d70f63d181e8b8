// MI355X (gfx950) bitHTM timestep engine: HIP kernels + the C ABI of include/bithtm_hip.h.
//
// One handle = all device state of one SpatialPooler + TemporalMemory pair.  A timestep is a
// fixed sequence of kernel launches on one stream with NO host synchronisation: every
// data-dependent size (segments, matching segments, winners, work items) lives in a device
// counter block and kernels grid-stride over those counters.
//
// Semantics follow the reference lines cited next to each kernel (paths relative to the
// reference checkout) under the deterministic policies of DESIGN.md.  Compile with
// -ffp-contract=off: several kernels must round exactly like the NumPy expressions they replace.
//
// Internal cell encoding: enc = column * 32 + cell ("one 32-bit word per column" bitmaps);
// the ABI converts to / from the reference's flat id column * cell_dim + cell.

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/bithtm_hip.h"
#include "htm_fexp.h"
#include "htm_rng.h"

typedef unsigned long long u64;

#define SEL_MAX_PASSES 6
#define SEL_DIGIT 12
#define SEL_BINS 4096         // 1 << SEL_DIGIT
#define HIST_REP 8            // copies of the digit-0 histogram (its few hot bins take one atomic per block)
#define RB 512                // threads per block of the role kernels (overlap, select, learn, scan)
#define SCAN_SEGS 64          // segments per 256-thread block iteration of the segment scan
#define DEAD_CAP 256          // newly dead segment ids one rank can report per exchange
#define CAND_CAP 256          // growth candidates staged per wave
#define MAX_SLOTS 512
#define EPS32 1e-8f           // `epsilon=1e-8` against float32 arrays (weak Python scalar)

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
    int32_t n_active_cells;
    int32_t n_work;           // learning / punish work items of this step (front of the work array)
    int32_t n_bind;           // newly bound segments of this step (back of the work array, growing down)
    int32_t n_work_last;      // ... of the last completed step (telemetry)
    int32_t sel_fallbacks;    // steps whose top-k select took the in-kernel fallback (telemetry)
    uint32_t emit_epoch;      // bumped by every overlap launch: tags the records k_sp_emit's blocks exchange
    int32_t n_un;             // winners needing a new segment
    int32_t n_recycled, n_new, S_old;
    int32_t error;            // sticky capacity flags
    // Spatial Pooler select state, double-buffered by the parity of the step it belongs to (the
    // pipelined schedule computes step t+1's overlap / select digits while step t's TM runs)
    u64 sel_prefix[2];        // k-th largest key and how many of the keys equal to it are winners
    uint32_t sel_krem[2];
    u64 sel_pass_prefix[2][SEL_MAX_PASSES + 1];    // radix-select state entering pass p
    uint32_t sel_pass_krem[2][SEL_MAX_PASSES + 1];
};

struct Dev {
    int I, W, W4, Ipad, C, K, k, E, Scap, work_cap, sel_passes, colwords, low_zero, cand_d, cand_pairwise, cand_others;
    int world, c0, c1;        // this rank owns columns [c0, c1) (world == 1: everything)
    double sp_thr, sp_don, sp_doff;
    float coef, mom, dinc;
    double lrn_act, lrn_inact, pun_act, pun_inact;
    int lrn_prune, pun_prune;
    float perm_init, perm_thr;
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
    uint32_t *hist0;          // [2][HIST_REP][SEL_BINS]        digit 0: block b adds to copy b % HIST_REP
    uint32_t *sel_blk;        // [ceil(C/256)] packed (greater, equal) counts per 256-column block
    uint32_t *sel_rec;        // [ceil(C/256)][32] per-block bucket records of k_sp_emit (16 granules)
    int *active_cols[2];      // [k] ascending; parity double buffer (the pipelined schedule emits step t+1's
                              // list while step t's scan still reads its own)
    uint32_t *input_stage;    // [W] host-fed input
    // Temporal Memory
    uint32_t *act[2];         // [C] active-cell words, parity double buffer
    uint32_t *pred[2];        // [C] predicted-cell words
    uint32_t *win[2];         // [C] winner-cell words
    uint32_t *colbits[2];     // [ceil(C/64)*2] bitmap of the step's active columns
    int *winners[2];          // [k*32] winner cells (enc), ascending
    uint8_t *bursting;        // [k]
    uint32_t *winw_idx;       // [k] winner word of the idx-th active column (same as win[active_cols[idx]])
    uint8_t *actcnt;          // [k] popc(active word) of the idx-th active column
    uint32_t *unacc_word;     // [k]
    int *unacc_list;          // [k*32] winners without a matching segment, ascending
    int *seg_cell;            // [Scap] owning cell (enc)
    int *seg_nsyn;            // [Scap] valid synapses; rows are packed: slots [0, nsyn) are valid
    int *presyn;              // [Scap][E] presynaptic cell (enc)
    float *sperm;             // [Scap][E] float32 permanence
    int *segcount;            // [C*32] segments per cell
    uint32_t *cellmax;        // [C*32] float bits of max jittered potential per cell (0 = none)
    uint32_t *seg_info;       // [Scap] last scan: potential | activation << 12 | matching << 30 | active << 31
    float *seg_jit;           // [Scap] jittered potential of the matching segments
    uint32_t *work;           // [work_cap] segment | mode << 31 (0 = learn + grow, 1 = punish)
    int *recyc_cnt;           // [ceil(Scap/1024)] recyclable segments per 1024-segment block
    int *recyc_need;          // [2*k*32] (block, first rank) pairs of the blocks add_output draws from
    // column sharding (world > 1): speculative per-column words of ALL columns after the exchange,
    // and the ids of owned segments that fell below the matching threshold while learning
    uint32_t *spec_act, *spec_win, *spec_unacc, *spec_burst;      // [C] [C] [C] [ceil(C/32)]
    int *dead_list;           // [1 + DEAD_CAP]: count, ids
    Counters *ctr;
    unsigned long long *trace;    // [8][4096][2] BITHTM_TRACE=1: device clock at the start / end of every block of the
                                  // pipelined launches (slot = launch + 4 * step parity), else null
    uint32_t trace_until;         // ... of steps with an index below this (BITHTM_TRACE_UNTIL; default: all)
};

// ------------------------------------------------------------------------------------------
// small device helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }

template <int BS>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_wave, uint32_t &total) {
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) s_wave[wv] = x;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < BS / 64; ++i) {
        uint32_t t = s_wave[i];
        if (i < wv) woff += t;
        tot += t;
    }
    __syncthreads();
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
    base = __shfl(base, leader);
    return pred ? base + __popcll(m & lanemask_lt()) : -1;
}

// h[digit] += 1 for every lane with `active`, one LDS atomic per distinct digit in the wave (keys
// of neighbouring columns mostly share their leading digits: per-lane atomics would serialise).
// All lanes of the wave must call.
__device__ __forceinline__ void hist_add(uint32_t *h, uint32_t digit, bool active) {
    u64 todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t dl = __shfl(digit, leader);
        const u64 same = __ballot(active && digit == dl) & todo;
        if (lane_id() == leader) atomicAdd(&h[dl], (uint32_t)__popcll(same));
        todo &= ~same;
    }
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
__device__ __forceinline__ uint32_t enc_to_flat(int enc, int K) { return (uint32_t)((enc >> 5) * K + (enc & 31)); }

// ------------------------------------------------------------------------------------------
// Spatial Pooler

// projections.py:19 for whole rows (after htm_sp_set_permanence)
__global__ __launch_bounds__(256) void k_sp_build_mask(Dev d, int row_begin, int row_count) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int lane = lane_id();
    const int chunks = d.Ipad >> 6;                       // 64 elements per ballot
    for (long long t = wave; t < (long long)row_count * chunks; t += nwaves) {
        int row = row_begin + (int)(t / chunks), ch = (int)(t % chunks);
        int i = ch * 64 + lane;
        bool conn = (i < d.I) && (d.perm[(size_t)row * d.Ipad + i] >= d.sp_thr);
        u64 m = __ballot(conn);
        if (lane == 0) *(u64 *)&d.mask[(size_t)row * d.W + ch * 2] = m;
    }
}

// DenseProjection.process (projections.py:18-21) + ExponentialBoosting.process
// (regularizations.py:15-17).  G lanes share one row (G = power of two, W4 16-byte chunks per row).
// Sharded handles run it on their own rows only and leave the histogram to k_shard_unpack, which
// sees the keys of all columns.
// Roles are written against (blk, nblk) instead of blockIdx / gridDim so that two independent
// roles can share one launch (pipelined schedule: step t's TM work beside step t+1's SP work).
// sp = parity buffer of the SP step being computed; step_offset = that step minus the current one.
template <int BS>
__device__ __forceinline__ void role_overlap(const Dev &d, const uint32_t *__restrict__ bank, int n_inputs, int G,
                                             int p, int sp, int step_offset, int blk, int nblk, uint32_t *h) {
    const int gtid = blk * BS + threadIdx.x;
    const int nthreads = nblk * BS;
    const bool do_hist = d.world == 1;
    uint32_t *ghist = d.hist + sp * SEL_MAX_PASSES * SEL_BINS;
    if (gtid == 0) d.ctr->emit_epoch += 1;          // a new generation of k_sp_emit records
    if (do_hist) {
        for (int i = gtid; i < (d.sel_passes - 1) * SEL_BINS; i += nthreads) ghist[SEL_BINS + i] = 0;
        for (int i = threadIdx.x; i < SEL_BINS; i += BS) h[i] = 0;
        if (gtid == 0) {
            d.ctr->sel_pass_prefix[sp][0] = 0;
            d.ctr->sel_pass_krem[sp][0] = (uint32_t)d.k;
        }
        __syncthreads();
    }
    const uint4 *in4 = (const uint4 *)(bank + (size_t)((d.ctr->step[p] + (uint32_t)step_offset) % (uint32_t)n_inputs) * d.W);
    const uint4 *mask4 = (const uint4 *)d.mask;
    const int lane = lane_id();
    const int rpw = 64 / G, sub = lane / G, l = lane % G;
    const int wave = gtid >> 6, nwaves = nthreads >> 6;
    constexpr int U = 4;                           // row groups in flight per wave
    for (int row0 = d.c0 + wave * rpw * U; row0 < d.c1; row0 += nwaves * rpw * U) {
        int cnt[U];
        float dty[U];                              // fetched with the mask rows, not after the reduction
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cnt[u] = 0;
            const int row = row0 + u * rpw + sub;
            dty[u] = (l == 0 && row < d.c1) ? d.duty[row] : 0.f;
        }
        for (int j = l; j < d.W4; j += G) {
            const uint4 x = in4[j];
            uint4 m[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = row0 + u * rpw + sub;
                m[u] = row < d.c1 ? mask4[(size_t)row * d.W4 + j] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                cnt[u] += __popc(m[u].x & x.x) + __popc(m[u].y & x.y) + __popc(m[u].z & x.z) + __popc(m[u].w & x.w);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int cn = cnt[u];
            for (int o = G >> 1; o > 0; o >>= 1) cn += __shfl_xor(cn, o);
            const int row = row0 + u * rpw + sub;
            const bool owner = l == 0 && row < d.c1;
            u64 key = 0;
            if (owner) {
                d.overlap[sp][row] = cn;
                const float f = htm_exp_f32(d.coef * dty[u]);          // float32 product, documented exp
                const double bo = (double)f * (double)cn;              // exact (24-bit x <= 16-bit)
                d.boosted[sp][row] = bo;
                key = (u64)__double_as_longlong(bo);
                d.key[sp][row] = key;
            }
            if (do_hist) hist_add(h, (uint32_t)(key >> sel_shift(0)), owner);
        }
    }
    if (!do_hist) return;
    __syncthreads();
    uint32_t *g0 = d.hist0 + (size_t)(sp * HIST_REP + (blk & (HIST_REP - 1))) * SEL_BINS;
    for (int i = threadIdx.x; i < SEL_BINS; i += BS)
        if (h[i]) atomicAdd(&g0[i], h[i]);
}

__global__ __launch_bounds__(RB) void k_sp_overlap(Dev d, const uint32_t *__restrict__ bank, int n_inputs, int G, int p, int sp, int step_offset) {
    __shared__ uint32_t h[SEL_BINS];               // histogram of the top key digit (select pass 0)
    role_overlap<RB>(d, bank, n_inputs, G, p, sp, step_offset, blockIdx.x, gridDim.x, h);
}

// GlobalInhibition.process (regularizations.py:28-29) as an exact radix select of the k-th
// largest key, one 12-bit digit per launch.  There is no intra-kernel hand-off: every block of
// pass p re-derives the bucket chosen by pass p-1 from that pass's (complete) histogram.
//
// sel_resolve: given the state entering pass `prev` and its histogram, the state entering
// pass prev+1.  Called by all BS threads of the block; h is SEL_BINS words of LDS scratch.
template <int BS>
__device__ __forceinline__ void sel_resolve(const Dev &d, int sp, int prev, uint32_t *h, uint32_t *s_wave,
                                            u64 *out_prefix, uint32_t *out_krem, u64 *s_res_prefix, uint32_t *s_res_krem) {
    const int tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    const int shift = sel_shift(prev), nb = 1 << sel_bits(prev);
    const u64 prefix = d.ctr->sel_pass_prefix[sp][prev];
    const uint32_t krem = d.ctr->sel_pass_krem[sp][prev];
    const uint32_t *gh = d.hist + (sp * SEL_MAX_PASSES + prev) * SEL_BINS;
    constexpr int PER = SEL_BINS / BS;            // bins per thread, thread t owns [t*PER, (t+1)*PER)
    static_assert(PER % 4 == 0, "16-byte histogram accesses");
    {   // fetch the histogram with coalesced 16-byte loads, all in flight; regroup through LDS
        uint4 v[PER / 4];
#pragma unroll
        for (int j = 0; j < PER / 4; ++j) {
            const int b = 4 * (j * BS + tid);
            v[j] = (prev > 0 && b < nb) ? *(const uint4 *)(gh + b) : make_uint4(0, 0, 0, 0);
        }
        if (prev == 0) {                            // digit 0: sum the copies
            const uint32_t *g0 = d.hist0 + (size_t)sp * HIST_REP * SEL_BINS;
            for (int r = 0; r < HIST_REP; ++r)
#pragma unroll
                for (int j = 0; j < PER / 4; ++j) {
                    const uint4 a = *(const uint4 *)(g0 + (size_t)r * SEL_BINS + 4 * (j * BS + tid));
                    v[j].x += a.x; v[j].y += a.y; v[j].z += a.z; v[j].w += a.w;
                }
        }
#pragma unroll
        for (int j = 0; j < PER / 4; ++j) *(uint4 *)(h + 4 * (j * BS + tid)) = v[j];
    }
    __syncthreads();
    uint32_t cs = 0;
#pragma unroll
    for (int j = 0; j < PER / 4; ++j) {
        const uint4 v = *(const uint4 *)(h + tid * PER + 4 * j);
        cs += v.x + v.y + v.z + v.w;
    }
    uint32_t x = cs;                              // inclusive suffix sum inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_down(x, o);
        if (lane + o < 64) x += y;
    }
    if (lane == 0) s_wave[wv] = x;                // wave total
    __syncthreads();
    uint32_t above = x - cs;                      // keys in bins above my chunk, inside my wave ...
    for (int w = wv + 1; w < BS / 64; ++w) above += s_wave[w];      // ... plus the higher waves
    if (above < krem && krem <= above + cs) {     // exactly one thread
        for (int b = min((tid + 1) * PER, nb) - 1; b >= tid * PER; --b) {
            const uint32_t hb = h[b];
            if (above + hb >= krem) {
                *s_res_prefix = prefix | ((u64)b << shift);
                *s_res_krem = krem - above;
                break;
            }
            above += hb;
        }
    }
    __syncthreads();
    *out_prefix = *s_res_prefix;
    *out_krem = *s_res_krem;
}

struct SelShared { uint32_t h[SEL_BINS]; uint32_t wave[16]; u64 prefix; uint32_t krem; };

template <int BS>
__device__ __forceinline__ void role_sel_pass(const Dev &d, int pass, int sp, int blk, int nblk, SelShared *sh) {
    const int tid = threadIdx.x;
    u64 prefix;
    uint32_t krem;
    sel_resolve<BS>(d, sp, pass - 1, sh->h, sh->wave, &prefix, &krem, &sh->prefix, &sh->krem);
    if (blk == 0 && tid == 0) {
        d.ctr->sel_pass_prefix[sp][pass] = prefix;
        d.ctr->sel_pass_krem[sp][pass] = krem;
    }
    __syncthreads();
    const int shift = sel_shift(pass), bits = sel_bits(pass), nb = 1 << bits;
    const u64 himask = ~0ull << (shift + bits);
    for (int i = tid; i < nb; i += BS) sh->h[i] = 0;
    __syncthreads();
    const u64 *keys = d.key[sp];
    for (int c0 = blk * BS + (tid & ~63); c0 < d.C; c0 += nblk * BS) {
        const int c = c0 + lane_id();
        const u64 key = c < d.C ? keys[c] : 0;
        hist_add(sh->h, (uint32_t)(key >> shift) & (nb - 1), c < d.C && ((key ^ prefix) & himask) == 0);
    }
    __syncthreads();
    uint32_t *gh = d.hist + (sp * SEL_MAX_PASSES + pass) * SEL_BINS;
    for (int i = tid; i < nb; i += BS)
        if (sh->h[i]) atomicAdd(&gh[i], sh->h[i]);
}

__global__ __launch_bounds__(RB) void k_sel_pass(Dev d, int pass, int sp) {
    __shared__ SelShared sh;
    role_sel_pass<RB>(d, pass, sp, blockIdx.x, gridDim.x, &sh);
}

// per 256-column block: how many keys are above / equal to the k-th largest
__global__ __launch_bounds__(256) void k_sp_count(Dev d, int sp) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t h[SEL_BINS];
    __shared__ u64 s_prefix;
    __shared__ uint32_t s_krem;
    u64 T;
    uint32_t r;
    sel_resolve<256>(d, sp, d.sel_passes - 1, h, s_wave, &T, &r, &s_prefix, &s_krem);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        d.ctr->sel_prefix[sp] = T;              // skipped low digits are zero in every key
        d.ctr->sel_krem[sp] = r;
    }
    if (d.sel_passes > 1)                       // pass-0 histogram is consumed: clear it for its next use
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HIST_REP * SEL_BINS; i += gridDim.x * 256) d.hist0[(size_t)sp * HIST_REP * SEL_BINS + i] = 0;
    const int c = blockIdx.x * 256 + threadIdx.x;
    uint32_t v = 0;
    if (c < d.C) {
        u64 key = d.key[sp][c];
        v = (key > T) ? 1u : ((key == T) ? 0x10000u : 0u);
    }
    uint32_t total;
    block_excl_scan<256>(v, s_wave, total);
    if (threadIdx.x == 0) d.sel_blk[blockIdx.x] = total;
}

// TemporalMemory.process up to the winner cells (networks.py:95-104) for ONE active column,
// executed by a half-wave (lane j = cell j): bursting, best-matching cell (networks.py:73-82),
// least-used cell (:84-89).  idx = position of column a in the ascending active list.
struct ColumnWords { uint32_t act, winner, unacc; bool burst; };

// pw = prev_state.cell_prediction row of column a (0 when !col_ok)
__device__ __forceinline__ ColumnWords tm_column_words(const Dev &d, int p, int want_winner, bool col_ok, int a, uint32_t pw) {
    const int lane = lane_id(), half = lane >> 5, j = lane & 31;
    const bool valid = col_ok && j < d.K;
    const bool burst = pw == 0;
    const uint32_t act = burst ? cell_mask(d.K) : pw;        // networks.py:115
    const int has_distal = d.ctr->has_distal;
    float cm = -1.0f;
    if (valid && has_distal) cm = __uint_as_float(d.cellmax[a * 32 + j]);
    uint32_t winner = pw, unacc = 0;
    if (want_winner) {
        float colmax = cm;
        for (int o = 16; o > 0; o >>= 1) colmax = fmaxf(colmax, __shfl_xor(colmax, o));
        const bool col_matching = has_distal && colmax >= (float)d.match_thr;      // networks.py:80
        const bool best = valid && has_distal && fabsf(cm - colmax) < EPS32;       // :81
        float jit = 3.0e38f;
        if (valid) {
            uint32_t base = htm_stream_base(d.seed, HTM_STREAM_LEAST_USED, d.ctr->step[p]);
            jit = htm_jitter((float)d.segcount[a * 32 + j], htm_draw24(base, (uint32_t)(a * d.K + j), 0u));   // :86-87
        }
        float mn = jit;
        for (int o = 16; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
        const bool least = valid && fabsf(jit - mn) < EPS32;                       // :88
        const bool wbit = col_matching ? best : least;
        const u64 bw = __ballot(wbit);
        const uint32_t pick = (uint32_t)(bw >> (half * 32));
        if (burst) winner = pick;                                                  // :102
        const u64 bm = __ballot(valid && has_distal && !(cm < EPS32));             // cell has a matching segment
        unacc = has_distal ? (winner & ~(uint32_t)(bm >> (half * 32))) : 0u;       // projections.py:271
    }
    return ColumnWords{act, want_winner ? winner : 0u, unacc, burst};
}

// store the words of active column a, the idx-th of the ascending active list
__device__ __forceinline__ void tm_store_column(const Dev &d, int p, bool col_ok, int a, int idx, const ColumnWords &w) {
    if (col_ok && (lane_id() & 31) == 0) {
        d.act[p][a] = w.act;
        d.win[p][a] = w.winner;
        d.bursting[idx] = w.burst ? 1 : 0;
        d.unacc_word[idx] = w.unacc;
        d.winw_idx[idx] = w.winner;
        d.actcnt[idx] = (uint8_t)__popc(w.act);
    }
}

__device__ __forceinline__ void tm_activate_column(const Dev &d, int p, int want_winner, bool col_ok, int a, int idx, uint32_t pw) {
    if (d.world == 1) {
        tm_store_column(d, p, col_ok, a, idx, tm_column_words(d, p, want_winner, col_ok, a, pw));
    } else {          // the owner computed the words before the exchange
        ColumnWords w{0, 0, 0, false};
        if (col_ok) {
            w.act = d.spec_act[a];
            w.winner = want_winner ? d.spec_win[a] : 0u;
            w.unacc = want_winner ? d.spec_unacc[a] : 0u;
            w.burst = (d.spec_burst[a >> 5] >> (a & 31)) & 1u;
        }
        tm_store_column(d, p, col_ok, a, idx, w);
    }
}

// ---- column sharding: the two kernels around the exchange ---------------------------------
// wire format of one rank's record (oracle/sharded.py record_nbytes):
//   [boosted f64 x Cl][act u32 x Cl][win u32 x Cl][unacc u32 x Cl][bursting bits u32 x ceil(Cl/32)]
//   [n_dead u32][dead ids u32 x DEAD_CAP], padded to 16 bytes
__host__ __device__ __forceinline__ size_t shard_record_bytes(int cl) {
    size_t n = (size_t)cl * 20 + 4 * (size_t)((cl + 31) / 32) + 4 + 4 * DEAD_CAP;
    return (n + 15) / 16 * 16;
}

// before the exchange: what each OWN column would look like if it became active (this only
// needs the rank's own previous predictions, segment maxima and segment counts), its boosted
// overlap, and the segments that died during the previous step's learning
__global__ __launch_bounds__(256) void k_shard_pack(Dev d, int p, unsigned char *send) {
    const int cl = d.c1 - d.c0;
    double *r_boost = (double *)send;
    uint32_t *r_act = (uint32_t *)(send + (size_t)cl * 8);
    uint32_t *r_win = r_act + cl, *r_unacc = r_win + cl, *r_burst = r_unacc + cl;
    uint32_t *r_dead = r_burst + (cl + 31) / 32;
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 5;           // local column, one per half-wave
    const bool ok = i < cl;
    const int a = d.c0 + (ok ? i : 0);
    const ColumnWords w = tm_column_words(d, p, 1, ok, a, ok ? d.pred[p ^ 1][a] : 0u);
    // bursting bits: one 32-bit word per 32 columns = 16 consecutive waves' halves; use atomics
    if (ok && (lane_id() & 31) == 0) {
        r_boost[i] = d.boosted[p][a];
        r_act[i] = w.act;
        r_win[i] = w.winner;
        r_unacc[i] = w.unacc;
        if (w.burst) atomicOr(&r_burst[i >> 5], 1u << (i & 31));
    }
    if (blockIdx.x == 0) {
        const int n = min(d.dead_list[0], DEAD_CAP);
        if (threadIdx.x == 0) r_dead[0] = (uint32_t)n;
        for (int j = threadIdx.x; j < n; j += 256) r_dead[1 + j] = (uint32_t)d.dead_list[1 + j];
    }
}

__global__ __launch_bounds__(256) void k_shard_pack_clear(Dev d, unsigned char *send) {
    const int cl = d.c1 - d.c0;
    uint32_t *r_burst = (uint32_t *)(send + (size_t)cl * 20);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < (cl + 31) / 32; i += gridDim.x * 256) r_burst[i] = 0;
}

// after the exchange: the keys and speculative words of ALL columns in global column order, the
// histogram of the top key digit (what k_sp_overlap does on an unsharded handle), and the deaths
// the other ranks reported (only "fewer synapses than the matching threshold" matters here:
// projections.py:80)
__global__ __launch_bounds__(1024) void k_shard_unpack(Dev d, const unsigned char *recv, int rank, int sp) {
    __shared__ uint32_t h[SEL_BINS];
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nthreads = gridDim.x * blockDim.x;
    const int cl = d.c1 - d.c0;
    const size_t rb = shard_record_bytes(cl);
    uint32_t *ghist = d.hist + sp * SEL_MAX_PASSES * SEL_BINS;
    for (int i = gtid; i < (d.sel_passes - 1) * SEL_BINS; i += nthreads) ghist[SEL_BINS + i] = 0;
    for (int i = threadIdx.x; i < SEL_BINS; i += 1024) h[i] = 0;
    if (gtid == 0) {
        d.ctr->sel_pass_prefix[sp][0] = 0;
        d.ctr->sel_pass_krem[sp][0] = (uint32_t)d.k;
        d.dead_list[0] = 0;                        // reported; start collecting this step's
    }
    __syncthreads();
    for (int c0 = blockIdx.x * 1024 + (threadIdx.x & ~63); c0 < d.C; c0 += gridDim.x * 1024) {
        const int c = c0 + lane_id();
        u64 key = 0;
        if (c < d.C) {
            const int r = c / cl, i = c - r * cl;
            const unsigned char *rec = recv + (size_t)r * rb;
            const double bo = ((const double *)rec)[i];
            const uint32_t *r_act = (const uint32_t *)(rec + (size_t)cl * 8);
            key = (u64)__double_as_longlong(bo);
            d.boosted[sp][c] = bo;
            d.key[sp][c] = key;
            d.spec_act[c] = r_act[i];
            d.spec_win[c] = r_act[cl + i];
            d.spec_unacc[c] = r_act[2 * cl + i];
            if ((c & 31) == 0) {                   // cl is a multiple of 32: words do not straddle ranks
                d.spec_burst[c >> 5] = r_act[3 * cl + (i >> 5)];
            }
        }
        hist_add(h, (uint32_t)(key >> sel_shift(0)), c < d.C);
    }
    __syncthreads();
    uint32_t *g0 = d.hist0 + (size_t)(sp * HIST_REP + (blockIdx.x & (HIST_REP - 1))) * SEL_BINS;
    for (int i = threadIdx.x; i < SEL_BINS; i += 1024)
        if (h[i]) atomicAdd(&g0[i], h[i]);
    if (blockIdx.x == 0) {
        for (int r = 0; r < d.world; ++r) {
            if (r == rank) continue;
            const uint32_t *r_dead = (const uint32_t *)(recv + (size_t)r * rb + (size_t)cl * 20) + (cl + 31) / 32;
            const int n = min((int)r_dead[0], DEAD_CAP);
            for (int j = threadIdx.x; j < n; j += 1024) {
                const int seg = (int)r_dead[1 + j];
                d.seg_nsyn[seg] = 0;
                atomicAdd(&d.recyc_cnt[seg >> 10], 1);
            }
        }
    }
}

// ---- finishing the select inside k_sp_emit ---------------------------------------------------
// Two radix digits (24 key bits) are resolved by launches; after them the threshold bucket holds a
// handful of distinct keys (or one key many times, when overlaps tie).  Every 256-column block
// publishes ONE 128-byte record -- how many of its keys lie above the bucket, and its distinct
// bucket keys with multiplicities -- as self-validating 8-byte granules (write-through stores,
// L1-bypassing loads: MI355X guide, Guideline 16, form R2).  Every block reads all records, so each one
// derives the exact k-th key T, the number r of keys equal to T that win, and the winner counts
// of the blocks before it, without another launch.  A block with more than CAND_D distinct
// bucket keys (or more than CAND_MAX in total) switches ALL blocks, consistently, to an exact
// fallback: the remaining digits are resolved block-redundantly from the key array and the
// per-block counts are exchanged in a second tagged round.
#define CAND_D 8              // distinct bucket keys one block can publish
#define CAND_RAW 64           // ... and collect from its waves before merging duplicates
#define CAND_MAX 2048         // bucket entries a block can merge
#define CAND_OTHERS 160         // ... after folding the copies of one key, if at most this many others remain
#define CAND_PAIRWISE 160      // ... by comparing all pairs; above that, by radix refinement in LDS

// pick the bucket that contains the krem-th largest key of a histogram held in LDS
// (bins [0, nb)); all BS threads call; returns bucket and the keys above it
template <int BS>
__device__ __forceinline__ void sel_pick(const uint32_t *h, int nb, uint32_t krem, uint32_t *s_wave,
                                         uint32_t *s_out /*[2]*/, uint32_t *bucket, uint32_t *above_out) {
    const int tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    constexpr int PER = SEL_BINS / BS;
    uint32_t cs = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int b = tid * PER + j;
        cs += b < nb ? h[b] : 0u;
    }
    uint32_t x = cs;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_down(x, o);
        if (lane + o < 64) x += y;
    }
    if (lane == 0) s_wave[wv] = x;
    __syncthreads();
    uint32_t above = x - cs;
    for (int w = wv + 1; w < BS / 64; ++w) above += s_wave[w];
    if (above < krem && krem <= above + cs) {
        for (int b = min((tid + 1) * PER, nb) - 1; b >= tid * PER; --b) {
            const uint32_t hb = h[b];
            if (above + hb >= krem) { s_out[0] = (uint32_t)b; s_out[1] = above; break; }
            above += hb;
        }
    }
    __syncthreads();
    *bucket = s_out[0];
    *above_out = s_out[1];
}

// Emit the winners in ascending column order (ties: lower index first), clear the dense per-column
// words of the non-winners and, as `mode` asks, update the duty cycle (regularizations.py:19-21,
// float32, two separately rounded operations: EMIT_DUTY) and run the Temporal Memory's per-column
// activation for the winners of this block (EMIT_ACTIVATE).  One block per 256 columns.  `fused`
// (grids of at most 1024 blocks, all co-resident): the select is finished here (above); otherwise
// T, r and the per-block counts come from k_sel_pass / k_sp_count launches.
// EMIT_CLEAR: also zero the dense per-column words of the step.  The pipelined schedule runs this
// role with mode 0 one step ahead, beside the previous step's learning (which still reads the
// words a clear would zero); see the pipelined schedule below.
#define EMIT_DUTY 1
#define EMIT_ACTIVATE 2
#define EMIT_CLEAR 4
#define EMIT_ALL 7
struct EmitShared {
    uint32_t h[SEL_BINS];
    u64 prefix, T;
    u64 bk[CAND_RAW];
    uint32_t bc[CAND_RAW];
    uint16_t ec[CAND_MAX], eb[CAND_MAX];
    uint32_t mh[256];
    u64 ok[CAND_OTHERS + 1];
    uint32_t oc[CAND_OTHERS + 1], c0;
    int n_others;
    uint32_t predw[256];
    int col[256];
    uint32_t wave[4];
    uint32_t gt, eq, out[2], flags, krem, r;
    int n, nraw, ne;
};

__device__ __forceinline__ void role_emit(const Dev &d, int p, int want_winner, int fused, int mode, int b, int nblk, EmitShared *sh) {
    uint32_t *h = sh->h;
    uint32_t *s_wave = sh->wave, *s_out = sh->out, *s_predw = sh->predw, *s_bc = sh->bc;
    u64 *s_bk = sh->bk;
    int *s_col = sh->col;
    uint32_t &s_gt = sh->gt, &s_eq = sh->eq, &s_flags = sh->flags, &s_krem = sh->krem, &s_r = sh->r;
    u64 &s_prefix = sh->prefix, &s_T = sh->T;
    int &s_n = sh->n, &s_nraw = sh->nraw, &s_ne = sh->ne;
    // merged bucket entries live in the histogram's LDS once the launched digits are resolved
    u64 *s_ek = (u64 *)h;                           // [CAND_MAX] keys
    uint16_t *s_ec = sh->ec, *s_eb = sh->eb;        // [CAND_MAX] multiplicities (12 bits), publishing block
    uint32_t *s_mh = sh->mh;
    static_assert(2 * CAND_MAX <= SEL_BINS, "bucket keys must fit the histogram");
    const int tid = threadIdx.x, lane = lane_id();
    if (tid == 0) { s_gt = 0; s_eq = 0; s_n = 0; s_nraw = 0; s_ne = 0; s_flags = 0; }
    const int c = b * 256 + tid;
    // independent of everything below: in flight while the select state is resolved
    const u64 my_key = c < d.C ? d.key[p][c] : 0;
    const bool own_col = c < d.C && c >= d.c0 && c < d.c1;
    const float my_duty = (own_col && (mode & EMIT_DUTY)) ? d.duty[c] : 0.f;
    const bool tm_here = d.act[0] && (mode & EMIT_ACTIVATE);
    s_predw[tid] = (c < d.C && tm_here && d.world == 1) ? d.pred[p ^ 1][c] : 0u;
    u64 T;
    uint32_t r;                                     // how many of the keys == T are selected
    bool second_round = false;                      // per-block counts still to be exchanged
    const uint32_t epoch = (d.ctr->emit_epoch & 0x3FFu) + 1u;           // 1..1024, changes with every overlap launch
    if (fused) {
        u64 P;
        uint32_t krem;
        sel_resolve<256>(d, p, d.sel_passes - 1, h, s_wave, &P, &krem, &s_prefix, &s_krem);
        __syncthreads();
        const int lowbits = sel_shift(d.sel_passes - 1);        // key bits not resolved by launches
        const u64 hiP = P >> lowbits, hi = my_key >> lowbits;
        const bool c_gt = c < d.C && hi > hiP, c_cand = c < d.C && hi == hiP;
        // ---- this block's record
        {
            const u64 mg = __ballot(c_gt);
            if (lane == 0 && mg) atomicAdd(&s_gt, (uint32_t)__popcll(mg));        // s_gt: keys above the bucket, for now
            u64 todo = __ballot(c_cand);
            while (todo) {                           // group equal bucket keys inside the wave
                const int leader = __ffsll((long long)todo) - 1;
                const u64 kl = ((u64)__shfl((uint32_t)(my_key >> 32), leader) << 32) | __shfl((uint32_t)my_key, leader);
                const u64 same = __ballot(c_cand && my_key == kl) & todo;
                if (lane == leader) {
                    const int slot = atomicAdd(&s_nraw, 1);
                    if (slot < CAND_RAW) { s_bk[slot] = kl; s_bc[slot] = (uint32_t)__popcll(same); }
                }
                todo &= ~same;
            }
        }
        __syncthreads();
        const int nraw = min(s_nraw, CAND_RAW);
        const uint32_t my_gt_hi = s_gt;
        int first = -1;                              // merge duplicates that came from different waves
        uint32_t my_cnt = 0;
        if (tid < nraw) {
            my_cnt = s_bc[tid];
            for (first = 0; s_bk[first] != s_bk[tid]; ++first) {}
        }
        __syncthreads();
        if (tid < nraw && first != tid) atomicAdd(&s_bc[first], my_cnt);
        __syncthreads();
        // record = up to 8 self-validating 64-bit granules (form R2: every granule carries the epoch, one
        // aligned 8-byte write-through store each, so no separate tag and no drain):
        //   [0]      epoch:12 | overflow:1 | pairs:4 | keys above the bucket:9 | multiplicity:9 | key bits:29
        //   [j >= 1] epoch:12 | multiplicity:12 | low 40 key bits   (the high bits are the bucket's)
        // The first pair rides in the head granule -- the low_zero bottom bits of every key are zero, so
        // 29 bits hold the rest for input_dim up to 2^17 -- and a block with at most one bucket key, the
        // usual case and the one of a many-way tie, is read with a single load.
        u64 *rec = (u64 *)(d.sel_rec + (size_t)b * 32);
        const u64 etag = (u64)epoch << 52;
        const u64 lowmask = (1ull << lowbits) - 1ull;
        const bool inline_ok = lowbits - d.low_zero <= 29;
        if (tid < 64) {                              // wave 0 compacts the survivors into the record
            const bool alive = tid < nraw && first == tid;
            const u64 ma = __ballot(alive);
            const int n_pairs = __popcll(ma), pos = __popcll(ma & lanemask_lt());
            const bool overflow = s_nraw > CAND_RAW || n_pairs > d.cand_d || (n_pairs > 0 && !inline_ok);
            const u64 mine = alive ? (s_bk[tid] & lowmask) : 0ull;
            const uint32_t cnt = alive ? s_bc[tid] : 0u;
            if (alive && pos >= 1 && pos < CAND_D)
                __hip_atomic_store(rec + pos, etag | ((u64)cnt << 40) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int l0 = ma ? __ffsll((long long)ma) - 1 : 0;      // the lane of pair 0
            const u64 k0 = ((u64)__shfl((uint32_t)(mine >> 32), l0) << 32) | __shfl((uint32_t)mine, l0);
            const uint32_t c0 = __shfl(cnt, l0);
            if (tid == 0) {
                const u64 pair0 = (ma && inline_ok) ? (((u64)(c0 & 0x1FFu) << 29) | (k0 >> d.low_zero)) : 0ull;
                __hip_atomic_store(rec, etag | (overflow ? (1ull << 51) : 0ull) | ((u64)min(n_pairs, CAND_D) << 47) | ((u64)my_gt_hi << 38) | pair0,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (tid == 0) s_gt = 0;
        // ---- everybody's records: the head granule is polled alone (one lane-load per spin keeps the
        // polling traffic low); further pairs, if any, are fetched in one batch; each granule validates itself
        uint32_t gthi_before = 0;
        for (int rb = tid; rb < nblk; rb += 256) {
            const u64 *rr = (const u64 *)(d.sel_rec + (size_t)rb * 32);
            u64 g[CAND_D];
            for (int spins = 0;; ++spins) {
                g[0] = __hip_atomic_load(rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((g[0] >> 52) == epoch) break;
                if (spins >= (1 << 20)) { atomicOr(&d.ctr->error, 16); g[0] = 0; break; }      // a block never arrived
                __builtin_amdgcn_s_sleep(2);
            }
            int np = (g[0] >> 52) == epoch ? (int)((g[0] >> 47) & 0xFu) : 0;
            for (int spins = 0; np > 1; ++spins) {
#pragma unroll
                for (int j = 1; j < CAND_D; ++j) g[j] = __hip_atomic_load(rr + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool ok = true;
#pragma unroll
                for (int j = 1; j < CAND_D; ++j) ok = ok && (j >= np || (g[j] >> 52) == epoch);
                if (ok) break;
                if (spins >= (1 << 20)) { atomicOr(&d.ctr->error, 16); np = 0; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (rb < b) gthi_before += (uint32_t)((g[0] >> 38) & 0x1FFu);
            if ((g[0] >> 51) & 1ull) atomicOr(&s_flags, 1u);
#pragma unroll
            for (int j = 0; j < CAND_D; ++j)
                if (j < np) {
                    const int slot = atomicAdd(&s_ne, 1);
                    if (slot < CAND_MAX) {
                        const u64 low = j == 0 ? (g[0] & 0x1FFFFFFFull) << d.low_zero : g[j] & lowmask;
                        s_ek[slot] = (hiP << lowbits) | low;
                        s_ec[slot] = (uint16_t)(j == 0 ? (g[0] >> 29) & 0x1FFu : (g[j] >> 40) & 0xFFFu);
                        s_eb[slot] = (uint16_t)rb;
                    }
                }
        }
        __syncthreads();
        const int ne = s_ne;
        if (!(s_flags & 1u) && ne <= CAND_MAX) {
            bool folded = false;
            if (ne <= d.cand_pairwise) {                // the krem-th largest of the merged bucket: all pairs
                for (int e = tid; e < ne; e += 256) {
                    const u64 ke = s_ek[e];
                    uint32_t ng = 0, nq = 0;
                    for (int f = 0; f < ne; ++f) {
                        const u64 kf = s_ek[f];
                        const uint32_t cf = s_ec[f];
                        ng += kf > ke ? cf : 0u;
                        nq += kf == ke ? cf : 0u;
                    }
                    if (ng < krem && krem <= ng + nq) { s_T = ke; s_r = krem - ng; }
                }
                __syncthreads();
                T = s_T;
                r = s_r;
            } else {
                // many entries: overlaps tie and most blocks report the same key.  Fold the copies of the
                // first entry's key into one entry; if few others remain, all pairs again
                const u64 K0 = s_ek[0];
                if (tid == 0) { sh->n_others = 0; sh->c0 = 0; }
                __syncthreads();
                uint32_t c0 = 0;
                for (int e = tid; e < ne; e += 256) {
                    const u64 ke = s_ek[e];
                    if (ke == K0) {
                        c0 += s_ec[e];
                    } else {
                        const int pos = atomicAdd(&sh->n_others, 1);
                        if (pos < CAND_OTHERS) { sh->ok[pos] = ke; sh->oc[pos] = s_ec[e]; }
                    }
                }
                for (int o = 32; o > 0; o >>= 1) c0 += __shfl_xor(c0, o);
                if (lane == 0 && c0) atomicAdd(&sh->c0, c0);
                __syncthreads();
                const int no = sh->n_others;
                folded = no < d.cand_others;
                if (folded) {
                    if (tid == 0) { sh->ok[no] = K0; sh->oc[no] = sh->c0; }
                    __syncthreads();
                    for (int e = tid; e <= no; e += 256) {
                        const u64 ke = sh->ok[e];
                        uint32_t ng = 0, nq = 0;
                        for (int f = 0; f <= no; ++f) {
                            const u64 kf = sh->ok[f];
                            const uint32_t cf = sh->oc[f];
                            ng += kf > ke ? cf : 0u;
                            nq += kf == ke ? cf : 0u;
                        }
                        if (ng < krem && krem <= ng + nq) { s_T = ke; s_r = krem - ng; }
                    }
                    __syncthreads();
                    T = s_T;
                    r = s_r;
                }
            }
            if (ne > d.cand_pairwise && !folded) {    // still many distinct keys: 8-bit radix refinement
                u64 pref = 0;                         // over the entries, one bin per thread
                uint32_t rem = krem;
                for (int top = lowbits; top > d.low_zero;) {
                    const int bits = min(8, top), shift = top - bits, nb = 1 << bits;
                    s_mh[tid] = 0;
                    __syncthreads();
                    for (int e = tid; e < ne; e += 256) {
                        const u64 kl = s_ek[e] & lowmask;
                        if (((kl ^ pref) >> top) == 0) atomicAdd(&s_mh[(uint32_t)(kl >> shift) & (nb - 1)], (uint32_t)s_ec[e]);
                    }
                    __syncthreads();
                    const uint32_t rv = s_mh[255 - tid];      // bins from the top; bins >= nb are empty
                    uint32_t total;
                    const uint32_t above = block_excl_scan<256>(rv, s_wave, total);
                    if (rv && above < rem && rem <= above + rv) { s_out[0] = 255u - (uint32_t)tid; s_out[1] = above; }
                    __syncthreads();
                    pref |= (u64)s_out[0] << shift;
                    rem -= s_out[1];
                    top = shift;
                }
                T = (hiP << lowbits) | pref;
                r = rem;
            }
            uint32_t g = gthi_before, e2 = 0;         // winners of the blocks before this one
            for (int e = tid; e < ne; e += 256)
                if (s_eb[e] < b) {
                    if (s_ek[e] > T) g += s_ec[e];
                    else if (s_ek[e] == T) e2 += s_ec[e];
                }
            for (int o = 32; o > 0; o >>= 1) { g += __shfl_xor(g, o); e2 += __shfl_xor(e2, o); }
            if (lane == 0) { atomicAdd(&s_gt, g); atomicAdd(&s_eq, e2); }
        } else {
            // exact fallback: resolve the remaining digits from the key array, redundantly per block
            u64 P2 = P;
            uint32_t k2 = krem;
            const u64 *keys = d.key[p];
            for (int top = lowbits; top > d.low_zero;) {
                const int bits = min(SEL_DIGIT, top), shift = top - bits, nb = 1 << bits;
                for (int i = tid; i < nb; i += 256) h[i] = 0;
                __syncthreads();
                for (int c0 = (tid & ~63); c0 < d.C; c0 += 256) {
                    const int cc = c0 + lane;
                    const u64 kk = cc < d.C ? keys[cc] : 0;
                    hist_add(h, (uint32_t)(kk >> shift) & (nb - 1), cc < d.C && ((kk ^ P2) >> top) == 0);
                }
                __syncthreads();
                uint32_t bucket, above;
                sel_pick<256>(h, nb, k2, s_wave, s_out, &bucket, &above);
                P2 |= (u64)bucket << shift;
                k2 -= above;
                top = shift;
                __syncthreads();
            }
            T = P2;
            r = k2;
            second_round = true;
            if (b == 0 && tid == 0) d.ctr->sel_fallbacks += 1;
        }
        if (b == 0 && tid == 0) { d.ctr->sel_prefix[p] = T; d.ctr->sel_krem[p] = r; }
        // the pass-0 histogram is consumed: clear it for its next use (here, not earlier: a barrier
        // waits for outstanding stores, and the record exchange above is the critical chain)
        if (d.sel_passes > 1)
            for (int i = b * 256 + tid; i < HIST_REP * SEL_BINS; i += nblk * 256) d.hist0[(size_t)p * HIST_REP * SEL_BINS + i] = 0;
    } else {
        T = d.ctr->sel_prefix[p];
        r = d.ctr->sel_krem[p];
    }
    __syncthreads();
    uint32_t flag = 0;
    if (c < d.C) flag = (my_key > T) ? 1u : ((my_key == T) ? 0x10000u : 0u);
    uint32_t total;
    const uint32_t ex = block_excl_scan<256>(flag, s_wave, total);
    if (second_round || !fused) {
        uint32_t g = 0, e = 0;
        if (fused) {                                // tagged words, second round of this step
            const uint32_t tag2 = epoch | 0x800u;
            if (tid == 0)
                __hip_atomic_store(&d.sel_blk[b], (tag2 << 20) | ((total >> 16) << 10) | (total & 0xFFFFu),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int i = tid; i < b; i += 256) {
                uint32_t v = 0;
                int spins = 0;
                do {
                    v = __hip_atomic_load(&d.sel_blk[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((v >> 20) == tag2) break;
                    __builtin_amdgcn_s_sleep(1);
                } while (++spins < (1 << 22));
                if ((v >> 20) != tag2) atomicOr(&d.ctr->error, 16);
                g += v & 0x3FFu;
                e += (v >> 10) & 0x3FFu;
            }
        } else {
            for (int i = tid; i < b; i += 256) {
                const uint32_t v = d.sel_blk[i];
                g += v & 0xFFFFu;
                e += v >> 16;
            }
        }
        for (int o = 32; o > 0; o >>= 1) { g += __shfl_xor(g, o); e += __shfl_xor(e, o); }
        if (lane == 0) { atomicAdd(&s_gt, g); atomicAdd(&s_eq, e); }
    }
    __syncthreads();
    const uint32_t gt_before = s_gt, eq_before = s_eq;
    const uint32_t g_run = gt_before + (ex & 0xFFFFu), e_run = eq_before + (ex >> 16);
    const int first_pos = (int)(gt_before + min(eq_before, r));
    const bool sel_any = c < d.C && ((flag & 1u) || ((flag >> 16) && e_run < r));
    if (d.act[0]) {
        const u64 mcol = __ballot(sel_any);
        if (lane_id() == 0) *(u64 *)&d.colbits[p][(b * 256 + (tid & ~63)) >> 5] = mcol;
    }
    if (c < d.C) {
        const bool sel = sel_any;
        if (own_col && (mode & EMIT_DUTY)) {
            float dc = my_duty * d.mom;
            if (sel) dc = dc + d.dinc;
            d.duty[c] = dc;
        }
        if (sel) {
            const int pos = (int)(g_run + min(e_run, r));
            d.active_cols[p][pos] = c;
            s_col[pos - first_pos] = c;
            atomicAdd(&s_n, 1);
        }
        if (d.act[0] && (mode & EMIT_CLEAR)) {     // Temporal Memory present
            d.pred[p][c] = 0;
            if (!sel || !(mode & EMIT_ACTIVATE)) { d.act[p][c] = 0; d.win[p][c] = 0; }
        }
    }
    if (b == 0 && tid == 0 && !d.act[0]) d.ctr->step[p ^ 1] = d.ctr->step[p] + 1;   // SP-only handle
    if (!tm_here) return;
    __syncthreads();
    const int n_sel = s_n;
    for (int i0 = 0; i0 < n_sel; i0 += 8) {        // 8 half-waves
        const int i = i0 + (tid >> 5);
        const bool ok = i < n_sel;
        const int a = ok ? s_col[i] : 0;
        tm_activate_column(d, p, want_winner, ok, a, first_pos + i, ok ? s_predw[a - b * 256] : 0u);
    }
}

__global__ __launch_bounds__(256) void k_sp_emit(Dev d, int p, int want_winner, int fused, int mode) {
    __shared__ EmitShared sh;
    role_emit(d, p, want_winner, fused, mode, blockIdx.x, gridDim.x, &sh);
}

// DenseProjection.update (projections.py:23-24) on the k winner rows, fused with the rebuild
// of those rows' connected mask.  One block per winner row.
__global__ __launch_bounds__(256) void k_sp_learn(Dev d, const uint32_t *__restrict__ bank, int n_inputs, int p) {
    const uint32_t *in = bank + (size_t)(d.ctr->step[p] % (uint32_t)n_inputs) * d.W;
    const int row = d.active_cols[p][blockIdx.x];
    double *prow = d.perm + (size_t)row * d.Ipad;
    uint32_t *mrow = d.mask + (size_t)row * d.W;
    for (int i0 = 0; i0 < d.Ipad; i0 += 256) {
        int i = i0 + threadIdx.x;
        bool conn = false;
        if (i < d.I) {
            bool on = (in[i >> 5] >> (i & 31)) & 1u;
            double v = prow[i] + (on ? d.sp_don : d.sp_doff);
            prow[i] = v;
            conn = v >= d.sp_thr;
        }
        u64 m = __ballot(conn);
        if (lane_id() == 0 && i < d.Ipad) *(u64 *)&mrow[i >> 5] = m;
    }
}

// ------------------------------------------------------------------------------------------
// Temporal Memory

// stand-alone TM: take the active columns from the caller, clear the per-step words
__global__ __launch_bounds__(256) void k_tm_load_active(Dev d, int p, const int *cols, int n) {
    for (int c = blockIdx.x * 256 + threadIdx.x; c < d.C; c += gridDim.x * 256) {
        d.act[p][c] = 0;
        d.pred[p][c] = 0;
        d.win[p][c] = 0;
        if (c < n) d.active_cols[p][c] = cols[c];
        if (c < d.colwords) d.colbits[p][c] = 0;
    }
}

// stand-alone TM: per-column activation, one active column per half-wave
__global__ __launch_bounds__(256) void k_tm_activate(Dev d, int p, int n_active, int want_winner) {
    const int idx = (blockIdx.x * 256 + threadIdx.x) >> 5;
    const bool ok = idx < n_active;
    const int a = ok ? d.active_cols[p][idx] : 0;
    if (ok && (threadIdx.x & 31) == 0) atomicOr(&d.colbits[p][a >> 5], 1u << (a & 31));
    tm_activate_column(d, p, want_winner, ok, a, idx, ok ? d.pred[p ^ 1][a] : 0u);
}

// bind segment `seg` to winner cell `cell` (projections.py:275-281) and queue it for learning at
// work-list slot `pos`; rows are packed, so clearing a recycled row (projections.py:82-85) is nsyn = 0
// Sharded: every rank records the new owner (segment ids are global), but only the owner of a
// cell keeps that cell's segment count, and only the new owner queues the segment; the others
// note how many synapses it will grow (projections.py:114-127 on an empty row: min(sampling,
// previous winners)), which is all they ever need to know about it.
__device__ __forceinline__ bool col_is_local(const Dev &d, int cell) { const int col = cell >> 5; return col >= d.c0 && col < d.c1; }

__device__ __forceinline__ void tm_bind_segment(const Dev &d, int seg, int cell, bool recycled, int pos, int grown) {
    if (recycled) {
        const int old = d.seg_cell[seg];
        if (col_is_local(d, old)) atomicSub(&d.segcount[old], 1);
    }
    d.seg_cell[seg] = cell;
    if (col_is_local(d, cell)) {
        d.seg_nsyn[seg] = 0;
        atomicAdd(&d.segcount[cell], 1);
        if (pos < 0) pos = atomicAdd(&d.ctr->n_work, 1);
        if (pos < d.work_cap) d.work[pos] = (uint32_t)seg; else atomicOr(&d.ctr->error, 4);
    } else {
        d.seg_nsyn[seg] = grown;
    }
}

// DenseProjection.update (projections.py:23-24) on winner row ri, fused with the rebuild of that
// row's connected mask, by TPR threads (t = 0..TPR-1): two float64 per lane (16-byte accesses); a
// wave covers 128 consecutive elements = four mask words, assembled from the ballots of its even
// and odd elements
template <int TPR>
// p: parity of the step the rows belong to; ahead = 1 when that step's index is not published yet
// (the row update runs beside the previous step's scan): it is step[p ^ 1] + 1 then
__device__ __forceinline__ void role_sp_row(const Dev &d, int p, const uint32_t *__restrict__ bank, int n_inputs, int ahead, int ri, int t) {
    const uint32_t step = ahead ? d.ctr->step[p ^ 1] + 1u : d.ctr->step[p];
    const uint32_t *in = bank + (size_t)(step % (uint32_t)n_inputs) * d.W;
    const int row = d.active_cols[p][ri];
    if (row < d.c0 || row >= d.c1) return;          // another rank's column
    double *prow = d.perm + (size_t)row * d.Ipad;
    uint32_t *mrow = d.mask + (size_t)row * d.W;
    for (int i0 = 0; i0 < d.Ipad; i0 += 2 * TPR) {
        const int e0 = i0 + 2 * t;                   // Ipad is a multiple of 128: e0 + 1 < Ipad whenever e0 < Ipad
        bool c0 = false, c1 = false;
        if (e0 < d.Ipad) {
            double2 v = *(double2 *)(prow + e0);
            const uint32_t bits = in[e0 >> 5] >> (e0 & 31);
            if (e0 < d.I) { v.x = v.x + ((bits & 1u) ? d.sp_don : d.sp_doff); c0 = v.x >= d.sp_thr; }
            if (e0 + 1 < d.I) { v.y = v.y + ((bits & 2u) ? d.sp_don : d.sp_doff); c1 = v.y >= d.sp_thr; }
            *(double2 *)(prow + e0) = v;
        }
        const u64 b0 = __ballot(c0), b1 = __ballot(c1);
        const int base = i0 + 2 * (t & ~63);
        if (lane_id() == 0 && base < d.Ipad) {
            u64 *mw = (u64 *)&mrow[base >> 5];
            mw[0] = spread32((uint32_t)b0) | (spread32((uint32_t)b1) << 1);
            mw[1] = spread32((uint32_t)(b0 >> 32)) | (spread32((uint32_t)(b1 >> 32)) << 1);
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
template <int BS>
__device__ __forceinline__ void role_mid(const Dev &d, int p, int n_active, int want_winner, int learning, int blk, int n_cls) {
    Counters *c = d.ctr;
    __shared__ int s_cnt, s_base;
    if (blk > 0) {
        if (!learning || !c->has_distal) return;
        const int q = p ^ 1;
        const int n = c->S;          // ids at or above the S of the last scan still hold info == 0
        const int stride = n_cls * BS;
        for (int i0 = (blk - 1) * BS; i0 < n; i0 += stride) {
            const int seg = i0 + threadIdx.x;
            bool learn = false, punish = false;
            const uint32_t info = seg < n ? d.seg_info[seg] : 0u;
            const int cell = seg < n ? d.seg_cell[seg] : 0;         // fetched with the info word, not after it
            if (info & 0x40000000u) {
                const int col = cell >> 5, bit = cell & 31;
                const bool is_winner = (d.win[p][col] >> bit) & 1u;
                const bool unpred = !((d.pred[q][col] >> bit) & 1u);                         // :266
                const bool best = fabsf(d.seg_jit[seg] - __uint_as_float(d.cellmax[cell])) < EPS32;   // :267
                learn = is_winner && ((info >> 31) || (unpred && best));                     // :268
                punish = d.act[p][col] == 0;                                                 // :269
            }
            if (threadIdx.x == 0) s_cnt = 0;
            __syncthreads();
            const u64 ml = __ballot(learn), mp = __ballot(punish);
            const int n_l = __popcll(ml), n_p = __popcll(mp);
            int woff = 0;
            if (lane_id() == 0 && n_l + n_p) woff = atomicAdd(&s_cnt, n_l + n_p);
            woff = __shfl(woff, 0);
            __syncthreads();
            if (threadIdx.x == 0 && s_cnt) s_base = atomicAdd(&c->n_work, s_cnt);     // one reservation per block
            __syncthreads();
            const int base = s_base + woff;
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
    for (int base = 0; base < n_active; base += LPT * BS) {
        const int i0 = base + LPT * (int)threadIdx.x;
        uint32_t v[LPT], ww[LPT], uw[LPT];
        int a[LPT];
        uint32_t vsum = 0;
        if (i0 < n_active) {                       // 16-byte loads (the arrays are padded by 8 entries), masked below
            const int4 a0 = *(const int4 *)(d.active_cols[p] + i0), a1 = *(const int4 *)(d.active_cols[p] + i0 + 4);
            const uint4 w0 = *(const uint4 *)(d.winw_idx + i0), w1 = *(const uint4 *)(d.winw_idx + i0 + 4);
            const uint4 u0 = *(const uint4 *)(d.unacc_word + i0), u1 = *(const uint4 *)(d.unacc_word + i0 + 4);
            const u64 ac = *(const u64 *)(d.actcnt + i0);
            a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
            ww[0] = w0.x; ww[1] = w0.y; ww[2] = w0.z; ww[3] = w0.w; ww[4] = w1.x; ww[5] = w1.y; ww[6] = w1.z; ww[7] = w1.w;
            uw[0] = u0.x; uw[1] = u0.y; uw[2] = u0.z; uw[3] = u0.w; uw[4] = u1.x; uw[5] = u1.y; uw[6] = u1.z; uw[7] = u1.w;
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                const bool ok = i0 + j < n_active;
                if (!ok) { ww[j] = 0; uw[j] = 0; }
                n_cells += ok ? (uint32_t)((ac >> (8 * j)) & 0xFFu) : 0u;
            }
        } else {
#pragma unroll
            for (int j = 0; j < LPT; ++j) { a[j] = 0; ww[j] = 0; uw[j] = 0; }
        }
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            v[j] = want_winner ? (uint32_t)__popc(ww[j]) | ((uint32_t)__popc(uw[j]) << 16) : 0u;      // winners | needing a segment
            vsum += v[j];
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
                run += v[j];
            }
        }
        carry_w += total & 0xFFFFu;
        carry_u += total >> 16;
    }
    for (int o = 32; o > 0; o >>= 1) n_cells += __shfl_xor(n_cells, o);
    if (lane_id() == 0) atomicAdd(&s_cells, n_cells);
    __syncthreads();
    const int n_un = (learning && c->has_distal) ? (int)carry_u : 0;
    if (threadIdx.x == 0) {
        c->n_win[p] = want_winner ? (int)carry_w : 0;
        c->has_winner[p] = want_winner;
        c->n_un = n_un;
        c->n_active_cells = (int)s_cells;
        if (n_un == 0) c->n_bind = 0;
    }
    if (n_un == 0) return;
    uint32_t carry = 0;                       // recyclable segments seen so far
    for (int base = 0; base < nb; base += BS) {
        const int b = base + threadIdx.x;
        const uint32_t v = base == 0 ? recyc_first : (b < nb ? (uint32_t)d.recyc_cnt[b] : 0u);
        uint32_t total;
        const uint32_t ex = block_excl_scan<BS>(v, s_wave, total);
        if (v > 0 && carry + ex < (uint32_t)n_un) {
            const int slot = atomicAdd(&s_nneed, 1);
            d.recyc_need[2 * slot] = b;
            d.recyc_need[2 * slot + 1] = (int)(carry + ex);
        }
        carry += total;
        if (carry >= (uint32_t)n_un) break;
    }
    __syncthreads();
    const int n_r = min(n_un, (int)carry);
    int n_new = n_un - n_r;
    if (S + n_new > d.Scap) {
        if (threadIdx.x == 0) atomicOr(&c->error, 1);
        n_new = max(d.Scap - S, 0);
    }
    // the bound segments are queued from the back of the work array (no reservation to wait for);
    // sharded: only the binds to own cells are queued, each with its own reservation at the front
    const bool whole = d.world == 1;
    __syncthreads();
    const int wbase = d.work_cap - (n_r + n_new);
    const int n_w = c->has_winner[p ^ 1] ? c->n_win[p ^ 1] : -1;
    const int grown = n_w > 0 ? min(d.sample, n_w) : 0;
    const int n_need = s_nneed;
    for (int i = 0; i < n_need; ++i) {          // each needed 1024-block: rank its recyclable segments
        const int b = d.recyc_need[2 * i], off = d.recyc_need[2 * i + 1];
        constexpr int IPT = 1024 / BS;             // consecutive segments per thread
        uint32_t fl[IPT], cnt = 0;
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const int seg = b * 1024 + (int)threadIdx.x * IPT + j;
            fl[j] = (seg < S && d.seg_nsyn[seg] < d.match_thr) ? 1u : 0u;
            cnt += fl[j];
        }
        uint32_t total;
        int rank = off + (int)block_excl_scan<BS>(cnt, s_wave, total);
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const int seg = b * 1024 + (int)threadIdx.x * IPT + j;
            if (fl[j] && rank < n_r) tm_bind_segment(d, seg, d.unacc_list[rank], true, whole ? wbase + rank : -1, grown);
            rank += (int)fl[j];
        }
    }
    for (int i = threadIdx.x; i < n_new; i += BS)
        tm_bind_segment(d, S + i, d.unacc_list[n_r + i], false, whole ? wbase + n_r + i : -1, grown);
    if (threadIdx.x == 0) {
        c->n_recycled = n_r;
        c->n_new = n_new;
        c->n_bind = whole ? n_r + n_new : 0;
        c->S_old = S;
        c->S = S + n_new;
    }
}

// SparseProjection.update_permanence (projections.py:97-109) and add_edge (:111-161) for one
// work item per wave.  Permanences: float64 sum, float32 store, prune on the float64 value; the
// surviving synapses are re-packed to the front of the row.  Growth: the n_add previous winner
// cells with the smallest keyed priority that the segment does not have yet.
template <int EPL, int BS>
struct LearnShared { u64 cand[BS / 64][CAND_CAP]; int keep[BS / 64][EPL * 64]; };

template <int EPL, int BS>
__device__ __forceinline__ void role_learn(const Dev &d, int p, int blk, int nblk, LearnShared<EPL, BS> *sh) {
    int (*s_keep)[EPL * 64] = sh->keep;
    u64 (*s_cand)[CAND_CAP] = sh->cand;
    Counters *c = d.ctr;
    {   // forget the previous scan's per-cell maxima (sparse clear; every reader ran in an earlier
        // launch) and reset what the coming scan accumulates
        const int n = c->has_distal ? c->S : 0;
        for (int i = blk * BS + threadIdx.x; i < n; i += nblk * BS)
            if (d.seg_info[i] & 0x40000000u) d.cellmax[d.seg_cell[i]] = 0u;
        const int nb = (c->S + 1023) >> 10;
        for (int i = blk * BS + threadIdx.x; i < nb; i += nblk * BS) d.recyc_cnt[i] = 0;
    }
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const int n_front = min(c->n_work, d.work_cap), n_back = c->n_bind;
    if (n_front + n_back > d.work_cap && blk == 0 && threadIdx.x == 0) atomicOr(&c->error, 4);
    const int n_work = min(n_front + n_back, d.work_cap);
    const uint32_t *act_prev = d.act[p ^ 1];
    const int *winners = d.winners[p ^ 1];
    const int n_w = c->has_winner[p ^ 1] ? c->n_win[p ^ 1] : -1;       // -1: winner_input is None
    const uint32_t base2 = htm_stream_base(d.seed, HTM_STREAM_GROWTH, c->step[p]);
    for (int item = blk * (BS / 64) + wv; item < n_work; item += nblk * (BS / 64)) {
        const uint32_t w = d.work[item < n_front ? item : d.work_cap - n_back + (item - n_front)];
        const int seg = (int)(w & 0x7FFFFFFFu), mode = (int)(w >> 31);
        const double dA = mode ? d.pun_act : d.lrn_act, dI = mode ? d.pun_inact : d.lrn_inact;
        const bool prune = mode ? d.pun_prune : d.lrn_prune;
        const int n = d.seg_nsyn[seg];
        int *prow = d.presyn + (size_t)seg * d.E;
        float *mrow = d.sperm + (size_t)seg * d.E;
        int n_keep = 0, n_active = 0;
#pragma unroll
        for (int jj = 0; jj < EPL; ++jj) {
            const int idx = jj * 64 + lane;
            const bool valid = idx < n;
            int ps = 0;
            float pm = 0.f;
            if (valid) { ps = prow[idx]; pm = mrow[idx]; }
            const bool a = valid && ((act_prev[ps >> 5] >> (ps & 31)) & 1u);
            const double p64 = (double)pm + (a ? dA : dI);               // :102-103
            const bool keep = valid && !(prune && p64 < 0.0);            // :105-108
            const u64 mk = __ballot(keep);
            if (keep) {
                const int pos = n_keep + __popcll(mk & lanemask_lt());
                prow[pos] = ps;
                mrow[pos] = (float)p64;                                   // :104
                s_keep[wv][pos] = ps;
            }
            n_keep += __popcll(mk);
            n_active += __popcll(__ballot(keep && a));                    // :114
        }
        __builtin_amdgcn_wave_barrier();
        int n_total = n_keep;
        if (mode == 0 && n_w > 0) {
            const int n_add = min(max(d.sample - n_active, 0), min(d.sample, n_w));     // :115
            if (n_add > 0) {
                // threshold T with n_add <= |{absent winners with priority < T}| <= CAND_CAP
                uint32_t lo = 0, hi = 1u << 24, T = 1u << 24;
                if (n_w > CAND_CAP) {
                    u64 est = ((u64)(2 * n_add + 16) << 24) / (u64)max(n_w - n_active, 1);
                    T = (uint32_t)min(est, (u64)(1u << 24));
                }
                // Each try stages every winner with priority < T, then drops the ones the segment already
                // has: one lane per staged winner walks the kept synapses once.  (Testing membership
                // inside the scan of the winner list made the whole wave walk them in every 64-winner
                // chunk with a hit: 30 us for a full row.)
                int found = 0;
                for (int iter = 0; iter < 64; ++iter) {
                    int staged = 0;
                    for (int b0 = 0; b0 < n_w; b0 += 64) {
                        const int i = b0 + lane;
                        uint32_t pr = 0;
                        bool take = false;
                        if (i < n_w) {
                            pr = htm_draw24(base2, (uint32_t)seg, enc_to_flat(winners[i], d.K));     // :120
                            take = pr < T;
                        }
                        const u64 mt = __ballot(take);
                        if (take) {
                            const int pos = staged + __popcll(mt & lanemask_lt());
                            if (pos < CAND_CAP) s_cand[wv][pos] = ((u64)pr << 32) | (uint32_t)i;
                        }
                        staged += __popcll(mt);
                    }
                    if (staged > CAND_CAP) {                      // too many for the staging area: lower T
                        hi = T;
                        if (hi - lo <= 1) { atomicOr(&c->error, 4); found = 0; break; }
                        T = (lo + hi) / 2;
                        continue;
                    }
                    __builtin_amdgcn_wave_barrier();
                    found = 0;
                    for (int e0 = 0; e0 < staged; e0 += 64) {     // :121-123, compacting in place (pos <= e)
                        const int e = e0 + lane;
                        u64 key = 0;
                        bool absent = false;
                        if (e < staged) {
                            key = s_cand[wv][e];
                            const int cell = winners[(uint32_t)key];
                            absent = true;
                            for (int qq = 0; qq < n_keep; ++qq)
                                if (s_keep[wv][qq] == cell) { absent = false; break; }
                        }
                        const u64 ma = __ballot(absent);
                        __builtin_amdgcn_wave_barrier();
                        if (absent) s_cand[wv][found + __popcll(ma & lanemask_lt())] = key;
                        found += __popcll(ma);
                    }
                    if (found >= n_add || T == (1u << 24)) break; // enough, or fewer absent winners than n_add: take all
                    lo = T;
                    T = (hi == (1u << 24)) ? (uint32_t)min((u64)T * 4u + 16u, (u64)hi) : (lo + hi + 1) / 2;
                }
                __builtin_amdgcn_wave_barrier();
                const int n_c = min(found, CAND_CAP), take_n = min(n_add, n_c);        // :125-127
                for (int e = lane; e < n_c; e += 64) {
                    const u64 key = s_cand[wv][e];
                    int rank = 0;
                    for (int f = 0; f < n_c; ++f) rank += s_cand[wv][f] < key;
                    if (rank < take_n) {
                        const int slot = n_keep + rank;
                        if (slot < d.E) {
                            prow[slot] = winners[(uint32_t)key];
                            mrow[slot] = d.perm_init;                                   // :149,158
                        } else {
                            atomicOr(&c->error, 2);
                        }
                    }
                }
                n_total = min(n_keep + take_n, d.E);                                    // :161
            }
        }
        if (lane == 0) {
            d.seg_nsyn[seg] = n_total;
            if (d.world > 1 && n >= d.match_thr && n_total < d.match_thr) {       // tell the other ranks
                const int slot = atomicAdd(&d.dead_list[0], 1);
                if (slot < DEAD_CAP) d.dead_list[1 + slot] = seg; else atomicOr(&c->error, 8);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];

template <int EPL>
__global__ __launch_bounds__(RB) void k_tm_learn(Dev d, int p) {
    role_learn<EPL, RB>(d, p, blockIdx.x, gridDim.x, (LearnShared<EPL, RB> *)dyn_lds);
}

// PredictiveProjection.process (projections.py:245-255): per segment, potential = active
// presynaptic cells; matching segments additionally count connected active synapses;
// per-cell prediction and max jittered potential (:229-239).  8 lanes per segment, 16-byte
// loads of the packed row, two segments in flight per lane group; a block owns SCAN_SEGS
// consecutive segment ids and also counts the recyclable ones among them (per 1024 ids) for the
// next step's add_output.  The last duty of a timestep: publish the next step index.
// use_lds: the bitmap of active columns is staged in LDS and consulted first, so that only the
// ~2 % of synapses whose presynaptic column is active touch the per-column cell words in L2.
// Branch-free: lanes whose column is inactive read act[0] instead (one shared cache line), so all
// LDS reads and then all global reads of a lane can be in flight together.
// use_lds: the bitmap of active columns is consulted in LDS first; only the ~2 % of synapses whose
// presynaptic column is active then read that column's cell word (lanes of inactive columns read
// act[0], one shared cache line, so the access stays branch-free).  Measured alternatives: a
// global gather for every synapse moves 64 B per bit; an LDS-only lookup (bitmap + prefix counts +
// active words) costs three bank-conflicted LDS reads per synapse and was 1.6x slower.
struct ScanLds { const uint32_t *colbits; };
__device__ __forceinline__ uint32_t scan_cell_active(const uint32_t *act, const ScanLds &L, int enc, bool valid, bool use_lds) {
    const int col = enc >> 5;
    uint32_t maybe = valid ? 1u : 0u;
    if (use_lds) maybe &= (L.colbits[col >> 5] >> (col & 31));
    uint32_t aw = 0;
    if (maybe) aw = act[col];          // exec-masked: only lanes of active columns issue a request
    return maybe & (aw >> (enc & 31));
}

// LDS: word 0 = recyclable counter; from word 4: column bitmap [colwords]
template <int BS, bool use_lds>
__device__ __forceinline__ void role_scan(const Dev &d, int p, int blk, int nblk, int n_spec, uint32_t *lds) {
    constexpr int SEGS = BS / 4;                   // segments per block iteration: BS/8 lane groups x 2 in flight
    int &s_recyc = *(int *)lds;
    uint32_t *s_colbits = lds + 4;
    const ScanLds L{s_colbits};
    Counters *c = d.ctr;
    const int S = c->S;
    if (blk == 0 && threadIdx.x == 0) {
        c->step[p ^ 1] = c->step[p] + 1;
        c->has_distal = 1;
        c->n_work_last = c->n_work + c->n_bind;
        c->n_work = 0;
        c->n_bind = 0;
    }
    const uint32_t *act = d.act[p];
    const uint32_t base3 = htm_stream_base(d.seed, HTM_STREAM_SEGMENT_JITTER, c->step[p]);
    // 8 lanes per segment, 16 bytes per lane: one 128-byte chunk = 32 synapse slots.  Packed rows
    // rarely exceed one chunk (growth tops a segment up to 32 active synapses), so a typical row
    // costs exactly 128 bytes of presynaptic ids.
    const int g = threadIdx.x >> 3, l = threadIdx.x & 7;
    constexpr int NG = BS / 8;                     // lane groups per block
    constexpr int U = SEGS / NG;                   // segments in flight per lane group
    bool staged = false;
    // In the first n_spec blocks (the ones that had segments when the host last saw the segment count)
    // the loads of the first batch do not wait for the count: rows up to the pool's capacity exist, so they
    // are fetched for ids clamped to it and masked once S has arrived: one dependent round trip less.
    for (int b = blk;; b += nblk) {
        const bool speculative = b == blk && blk < n_spec;
        if (!speculative && b * SEGS >= S) break;
        int seg[U], n[U], pot[U], conn[U], n_true[U], cellu[U];
        u64 bits[U];
        int4 ps[U], ps2[U];
        bool mine[U];
        // round trip 1: synapse count, owner cell and the first chunk of each row, all unconditional
        // (rows of other ranks' segments exist in the replicated address space; they are masked below)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            seg[u] = min(b * SEGS + u * NG + g, d.Scap - 1);
            n[u] = d.seg_nsyn[seg[u]];
            cellu[u] = d.seg_cell[seg[u]];
            ps[u] = *(const int4 *)(d.presyn + (size_t)seg[u] * d.E + l * 4);
        }
        if (!staged) {                               // the bitmap staging overlaps with those loads
            if (use_lds)
                for (int i = threadIdx.x; i < d.colwords; i += BS) s_colbits[i] = d.colbits[p][i];
            staged = true;
        }
        if (threadIdx.x == 0) s_recyc = 0;
        __syncthreads();
        if (speculative && b * SEGS >= S) break;     // (uniform in the block)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = b * SEGS + u * NG + g < S;
            mine[u] = d.world == 1 || col_is_local(d, cellu[u]);
            n_true[u] = ok ? n[u] : 0x7FFFFFFF;      // for the recyclable count (all ranks, all segments)
            if (!ok || !mine[u]) n[u] = 0;
            seg[u] = ok ? seg[u] : S;
        }
        // round trip 2 (only rows longer than one chunk): second chunk, in flight during the lookups of the first
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ps2[u] = make_int4(0, 0, 0, 0);
            if (n[u] > 32) ps2[u] = *(const int4 *)(d.presyn + (size_t)seg[u] * d.E + 32 + l * 4);
        }
        // chunk 1: all LDS lookups, then all cell-word reads, each as one batch
        {
            int e[U][4];
            uint32_t on[U][4], aw[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) { e[u][0] = ps[u].x; e[u][1] = ps[u].y; e[u][2] = ps[u].z; e[u][3] = ps[u].w; }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int col = e[u][qq] >> 5;
                    on[u][qq] = use_lds ? (s_colbits[col >> 5] >> (col & 31)) & 1u : 1u;
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    if (l * 4 + qq >= n[u]) on[u][qq] = 0;
                    aw[u][qq] = act[on[u][qq] ? (e[u][qq] >> 5) : 0];     // inactive columns: one shared line
                }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                u64 bb = 0;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) bb |= (u64)(on[u][qq] & (aw[u][qq] >> (e[u][qq] & 31))) << qq;
                bits[u] = bb;
            }
        }
        // chunk 2, same shape (skipped by waves in which no row is that long)
        bool any_long = false;
#pragma unroll
        for (int u = 0; u < U; ++u) any_long |= n[u] > 32;
        if (__any(any_long)) {
            int e[U][4];
            uint32_t on[U][4], aw[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) { e[u][0] = ps2[u].x; e[u][1] = ps2[u].y; e[u][2] = ps2[u].z; e[u][3] = ps2[u].w; }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int col = e[u][qq] >> 5;
                    on[u][qq] = use_lds ? (s_colbits[col >> 5] >> (col & 31)) & 1u : 1u;
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    if (32 + l * 4 + qq >= n[u]) on[u][qq] = 0;
                    aw[u][qq] = act[on[u][qq] ? (e[u][qq] >> 5) : 0];
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) bits[u] |= (u64)(on[u][qq] & (aw[u][qq] >> (e[u][qq] & 31))) << (4 + qq);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (n[u] > 64) {                         // rare: rows longer than two chunks
                const int *prow = d.presyn + (size_t)seg[u] * d.E;
                for (int i = 64 + l * 4, ch = 2; i < n[u]; i += 32, ++ch) {
                    const int4 pv = *(const int4 *)(prow + i);
                    const int e[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
                        bits[u] |= (u64)scan_cell_active(act, L, e[qq], i + qq < n[u], use_lds) << (ch * 4 + qq);
                }
            }
            int v = __popcll(bits[u]);
            for (int o = 4; o > 0; o >>= 1) v += __shfl_xor(v, o);
            pot[u] = v;
        }
        {   // connected active synapses of the matching segments (:171-172): the permanences of the first two
            // chunks of every matching row are fetched in one batch (one round trip, not one per chunk and row)
            float4 pm[U][2];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool matching = pot[u] >= d.match_thr;                         // :247
                const float *mrow = d.sperm + (size_t)seg[u] * d.E;
#pragma unroll
                for (int ch = 0; ch < 2; ++ch)
                    pm[u][ch] = (matching && ch * 32 + l * 4 < n[u]) ? *(const float4 *)(mrow + ch * 32 + l * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int cn = 0;
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const float e[4] = {pm[u][ch].x, pm[u][ch].y, pm[u][ch].z, pm[u][ch].w};
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
                        cn += ((bits[u] >> (ch * 4 + qq)) & 1ull) && (e[qq] >= d.perm_thr);
                }
                if (pot[u] >= d.match_thr && n[u] > 64) {                            // rare: longer rows
                    const float *mrow = d.sperm + (size_t)seg[u] * d.E;
                    for (int i = 64 + l * 4, ch = 2; i < n[u]; i += 32, ++ch) {
                        const float4 pv = *(const float4 *)(mrow + i);
                        const float e[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq)
                            cn += ((bits[u] >> (ch * 4 + qq)) & 1ull) && (e[qq] >= d.perm_thr);
                    }
                }
                for (int o = 4; o > 0; o >>= 1) cn += __shfl_xor(cn, o);
                conn[u] = cn;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (l == 0 && seg[u] < S) {
                if (n_true[u] < d.match_thr) atomicAdd(&s_recyc, 1);
                if (!mine[u]) continue;
                const bool matching = pot[u] >= d.match_thr;
                uint32_t info = (uint32_t)pot[u];
                if (matching) {
                    const bool active = conn[u] >= d.act_thr;                         // :250
                    const int cell = cellu[u];
                    const float jit = htm_jitter((float)pot[u], htm_draw24(base3, (uint32_t)seg[u], 0u));   // :234-235
                    atomicMax(&d.cellmax[cell], __float_as_uint(jit));               // :237
                    if (active) atomicOr(&d.pred[p][cell >> 5], 1u << (cell & 31));   // :251, networks.py:122
                    info |= ((uint32_t)conn[u] << 12) | 0x40000000u | (active ? 0x80000000u : 0u);
                    d.seg_jit[seg[u]] = jit;
                }
                d.seg_info[seg[u]] = info;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0 && s_recyc) atomicAdd(&d.recyc_cnt[(b * SEGS) >> 10], s_recyc);
    }
}

// use_lds is a compile-time switch: as a run-time flag it put a branch and a wait around every
// single LDS lookup, which serialised them
// MINW = 6 caps the kernel at 80 registers so that 6 blocks fit a CU and a pool of up to ~98 k
// segments is scanned by blocks that are all resident at once (the latency-bound regime of the bench
// workload); large pools are bandwidth-bound and run faster without the cap (MINW = 1).
template <bool use_lds, int MINW>
__global__ __launch_bounds__(256, MINW) void k_tm_scan(Dev d, int p, int n_spec) {
    role_scan<256, use_lds>(d, p, blockIdx.x, gridDim.x, n_spec, (uint32_t *)dyn_lds);
}

// ---- pipelined schedule: roles of different steps share every launch ---------------------------
// A forked stream / graph branch costs 17-29 us on this runtime and every dependent launch 1.2-3 us
// plus its own chain of memory round trips; heterogeneous blocks in one launch cost nothing.  The
// Spatial Pooler never reads Temporal Memory state, so inside a batched run it works ahead of the
// Temporal Memory, role by role, in the same four launches (t = the TM's step):
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
// overlap) sit beside the latency-bound mid and learn roles.  The look-ahead includes the SP's
// persistent updates (rows, duty cycle), so it only happens between two steps of one htm_run
// call: the last two steps of a run look ahead less (StepPlan) and no call returns with SP work
// outstanding.
struct TraceScope {                                 // BITHTM_TRACE=1: first / last device clock of every block
    unsigned long long *t;
    __device__ TraceScope(const Dev &d, int slot) {
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
    } else {                                       // one active column per half-wave
        const int idx = (((int)blockIdx.x - n_emit_blocks) * 256 + (int)threadIdx.x) >> 5;
        const bool ok = idx < n_active;
        const int a = ok ? d.active_cols[p][idx] : 0;
        tm_activate_column(d, p, 1, ok, a, idx, ok ? d.pred[p ^ 1][a] : 0u);
    }
}

// blocks [0, 1 + n_cls): the middle of the TM step; then one SP winner row per block -- of this step
// (rows_ahead = 0: one role per launch) or of the coming one (pipelined schedule) --; then the coming
// step's duty cycle (regularizations.py:19-21, float32, two roundings).  256-thread blocks: the
// dispatcher places them about five times faster, wave for wave, than 1024-thread ones (measured:
// 2000 small blocks start within 1 us, 800 large ones take 7), and all of them are resident at once.
__global__ __launch_bounds__(256) void k_mid_rows(Dev d, int p, int n_active, int want_winner, int learning, int n_cls,
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
    const int c = b * 256 + (int)threadIdx.x;
    if (b < n_duty_blocks && c < d.C) {
        float dc = d.duty[c] * d.mom;
        if ((d.colbits[q][c >> 5] >> (c & 31)) & 1u) dc = dc + d.dinc;
        d.duty[c] = dc;
    }
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
        const int c = b * 256 + (int)threadIdx.x, q = p ^ 1;
        if (c < d.C) {
            d.act[q][c] = 0;
            d.win[q][c] = 0;
            d.pred[q][c] = 0;
        }
        return;
    }
    b -= n_clear_blocks;
    role_scan<256, use_lds>(d, p, b, gridDim.x - n_sel_blocks - n_clear_blocks, n_spec, (uint32_t *)dyn_lds);
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
    if (threadIdx.x == 0) d.recyc_cnt[b] = s_recyc;
}

// ------------------------------------------------------------------------------------------
// host side

static thread_local std::string g_create_error;

struct htm_handle {
    htm_config cfg;
    Dev d;
    int device;
    hipStream_t stream;
    bool own_stream;
    long long step_host;
    std::string err;
    std::vector<void *> allocs;
    int *d_cols_stage;                    // stand-alone TM: active columns
    int rank, world;                      // column sharding
    const uint32_t *shard_bank;           // input of the step between htm_shard_begin and _finish
    int shard_n_inputs;
    bool shard_open;
    int G;                                // lanes per SP row
    int graph_steps;                      // steady-state steps per captured graph (BITHTM_GRAPH_STEPS)
    int seg_hint;                         // a lower bound of the segment count (see scan_spec_blocks)
    int *seg_pinned;                      // pinned word the end of each htm_run copies the count into
    int sp_blocks, sel_blocks, c256_blocks, s1024_blocks, scan_blocks;
    // graphs keyed by (parity, learning, bank, n_inputs)
    std::map<std::tuple<int, int, const void *, int>, hipGraphExec_t> graphs;
    // state import staging (htm_write of the MATCH_* / SEG_POTENTIAL fields, applied at commit)
    std::vector<int> imp_pot, imp_match_seg;
    std::vector<uint32_t> imp_match_info;
    std::vector<float> imp_match_jit;
    // profiling
    bool profile;
    hipEvent_t prof_last;                  // event closing the previous kernel of the profiled chain
    std::vector<hipEvent_t> prof_all;      // every event created (destroyed in htm_profile_read)
    std::vector<std::string> prof_names;
    std::vector<std::vector<std::pair<hipEvent_t, hipEvent_t>>> prof_events;
    std::vector<double> prof_ms;
    std::vector<long long> prof_n;
};

#define HIPCHK(h, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
            return HTM_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

template <typename T>
static int dalloc(htm_handle *h, T **p, size_t count) {
    void *q = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {
        h->err = "hipMalloc(" + std::to_string(bytes) + " bytes): " + hipGetErrorString(e);
        return HTM_ERR_HIP;
    }
    e = hipMemsetAsync(q, 0, bytes, h->stream);
    if (e != hipSuccess) { h->err = std::string("hipMemsetAsync: ") + hipGetErrorString(e); return HTM_ERR_HIP; }
    h->allocs.push_back(q);
    *p = (T *)q;
    return 0;
}

static int prof_slot(htm_handle *h, const char *name) {
    for (size_t i = 0; i < h->prof_names.size(); ++i)
        if (h->prof_names[i] == name) return (int)i;
    h->prof_names.push_back(name);
    h->prof_events.emplace_back();
    h->prof_ms.push_back(0.0);
    h->prof_n.push_back(0);
    return (int)h->prof_names.size() - 1;
}

// Profiling: hipExtLaunchKernelGGL stamps a start and a stop event with the kernel's own begin /
// end timestamps on the device -- the quantity rocprofv3's kernel trace reports -- so the two can
// be compared directly (an event recorded between launches would add its marker and the dependent
// launch gap to every kernel).
#define LAUNCH_ON(h, strm, shmem, name, kernel, grid, block, ...)                                 \
    do {                                                                                         \
        if ((h)->profile) {                                                                      \
            hipEvent_t e0_ = nullptr, e1_ = nullptr;                                             \
            hipEventCreate(&e0_);                                                                \
            hipEventCreate(&e1_);                                                                \
            hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), shmem, strm, e0_, e1_, 0, __VA_ARGS__); \
            (h)->prof_events[prof_slot(h, name)].push_back({e0_, e1_});                          \
        } else {                                                                                 \
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), shmem, strm, __VA_ARGS__);       \
        }                                                                                        \
    } while (0)
#define LAUNCH(h, name, kernel, grid, block, ...) LAUNCH_ON(h, (h)->stream, 0, name, kernel, grid, block, __VA_ARGS__)

static size_t learn_lds(int epl, int bs = RB) { return (size_t)(bs / 64) * CAND_CAP * 8 + (size_t)(bs / 64) * epl * 64 * 4; }
static int learn_epl(const Dev &d) { const int e = d.E / 64; return e <= 1 ? 1 : e == 2 ? 2 : e <= 4 ? 4 : 8; }
static size_t scan_lds(const Dev &d, int use_lds) { return 16 + (use_lds ? (size_t)d.colwords * 4 : 0); }
static const int kClassifyBlocks = 384;           // x 256 segments per pass of the learn / punish classification
static const int kLearnBlocks = 256;               // x RB/64 waves: one wave per learning / punished segment

static void launch_learn(htm_handle *h, int p) {
    Dev &d = h->d;
    const int epl = learn_epl(d);
    const size_t lds = learn_lds(epl);
    switch (epl) {
        case 1: LAUNCH_ON(h, h->stream, lds, "tm_learn", k_tm_learn<1>, kLearnBlocks, RB, d, p); break;
        case 2: LAUNCH_ON(h, h->stream, lds, "tm_learn", k_tm_learn<2>, kLearnBlocks, RB, d, p); break;
        case 4: LAUNCH_ON(h, h->stream, lds, "tm_learn", k_tm_learn<4>, kLearnBlocks, RB, d, p); break;
        default: LAUNCH_ON(h, h->stream, lds, "tm_learn", k_tm_learn<8>, kLearnBlocks, RB, d, p); break;
    }
}

// scan blocks that certainly have segments: the count only grows, and the host sees it now and then
// (htm_get_info, state import, and a copy queued at the end of every htm_run)
// (rounded down to a multiple of 64 blocks: the value is baked into captured graphs)
static int scan_spec_blocks(const htm_handle *h) { return std::min(h->seg_hint / SCAN_SEGS, h->scan_blocks) & ~63; }

// more segments than three rounds of resident blocks: the scan is bandwidth-bound (see k_tm_scan)
static bool scan_pool_is_large(const htm_handle *h) { return h->seg_hint > 3 * 1536 * SCAN_SEGS; }

static void launch_scan(htm_handle *h, int p, int use_lds) {
    Dev &d = h->d;
    const int spec = scan_spec_blocks(h);
    if (scan_pool_is_large(h)) {
        if (use_lds) LAUNCH_ON(h, h->stream, scan_lds(d, 1), "tm_scan", (k_tm_scan<true, 1>), h->scan_blocks, 256, d, p, spec);
        else LAUNCH_ON(h, h->stream, scan_lds(d, 0), "tm_scan", (k_tm_scan<false, 1>), h->scan_blocks, 256, d, p, spec);
    } else {
        if (use_lds) LAUNCH_ON(h, h->stream, scan_lds(d, 1), "tm_scan", (k_tm_scan<true, 6>), h->scan_blocks, 256, d, p, spec);
        else LAUNCH_ON(h, h->stream, scan_lds(d, 0), "tm_scan", (k_tm_scan<false, 6>), h->scan_blocks, 256, d, p, spec);
    }
}

// Front of SpatialPooler.process for the step with parity sp: overlap + boost (+ select digit 0)
// and the remaining select digits.
static void enqueue_sp_front(htm_handle *h, const uint32_t *bank, int n_inputs, int p) {
    Dev &d = h->d;
    LAUNCH(h, "sp_overlap", k_sp_overlap, h->sp_blocks, RB, d, bank, n_inputs, h->G, p, p, 0);
    for (int pass = 1; pass < d.sel_passes; ++pass) LAUNCH(h, "sp_select", k_sel_pass, h->sel_blocks, RB, d, pass, p);
}

// Rest of SpatialPooler.process: count + emit.  mode = EMIT_ALL: all of it, with the TM's per-column
// activation when the handle has a Temporal Memory.  sp_learn: the permanence update as a launch of
// its own (handles without a Temporal Memory, and the first step of a pipelined run).
static void enqueue_sp_back(htm_handle *h, const uint32_t *bank, int n_inputs, int p, int want_winner, int mode, bool sp_learn) {
    Dev &d = h->d;
    const int fused = h->c256_blocks <= 1024;      // all blocks co-resident: count inside emit
    if (!fused) LAUNCH(h, "sp_count", k_sp_count, h->c256_blocks, 256, d, p);
    LAUNCH(h, "sp_emit", k_sp_emit, h->c256_blocks, 256, d, p, want_winner, fused, mode);
    // (a forked graph branch for this independent update was measured at +17..29 us per step on
    // this runtime, against 2.3 us for one more kernel in the chain: tools/launch_overhead.hip)
    if (sp_learn) LAUNCH(h, "sp_learn", k_sp_learn, d.k, 256, d, bank, n_inputs, p);
}

// TemporalMemory.process after the per-column activation, one role per launch.  sp_rows: the SP
// permanence update of this step rides along in the middle launch.
static void enqueue_tm(htm_handle *h, int n_active, int learning, int want_winner, int p,
                       const uint32_t *bank, int n_inputs, bool sp_rows) {
    Dev &d = h->d;
    const int n_cls = learning ? kClassifyBlocks : 0;
    const int n_sp_rows = (sp_rows && learning && h->cfg.enable_sp) ? d.k : 0;
    LAUNCH(h, "tm_mid", k_mid_rows, 1 + n_cls + n_sp_rows, 256, d, p, n_active, want_winner, learning, n_cls, bank, n_inputs, n_sp_rows, 0, 0);
    launch_learn(h, p);
    launch_scan(h, p, scan_lds(d, 1) <= 64 * 1024);
}

// How a step is launched inside htm_run.
//   sp_done   the Spatial Pooler has already done this step (winner list, permanence rows, duty cycle)
//   next_sp   finish the SP's next step beside this step's TM (winner list, rows, duty cycle; its
//             overlap and select digits were computed one step earlier, or by the cold start)
//   next_front  compute overlap + select digits of the step after the next
// The look-ahead includes the SP's persistent updates, so it only ever happens between steps of one
// htm_run call (same bank, same learning flag): the last step of a run has neither, the one before it
// no next_front, and no call returns with SP work outstanding.
struct StepPlan { bool sp_done, next_sp, next_front; };

// the pipelined schedule needs the select finished inside one co-resident emit grid after two
// launched digits
static bool can_pipeline(const htm_handle *h) {
    return h->cfg.enable_sp && h->cfg.enable_tm && h->world == 1 && h->c256_blocks <= 1024 && h->d.sel_passes == 2;
}

// the four launches of a pipelined step (see the kernels): step p's Temporal Memory beside SP work of
// the following steps
static void enqueue_pipelined(htm_handle *h, int p, int learning, const uint32_t *bank, int n_inputs, StepPlan plan) {
    Dev &d = h->d;
    const int n_cls = learning ? kClassifyBlocks : 0;
    const int n_emit = plan.next_sp ? h->c256_blocks : 0;
    LAUNCH_ON(h, h->stream, sizeof(EmitShared), "tm_activate+sp_emit", k_open_emit, n_emit + (d.k * 32 + 255) / 256, 256, d, p, n_emit, d.k);
    const int n_rows = (plan.next_sp && learning) ? d.k : 0, n_duty = plan.next_sp ? h->c256_blocks : 0;
    LAUNCH(h, "tm_mid+sp_learn", k_mid_rows, 1 + n_cls + n_rows + n_duty, 256, d, p, d.k, 1, learning, n_cls, bank, n_inputs, n_rows, 1, n_duty);
    {
        const int epl = learn_epl(d);
        const size_t lds = std::max(learn_lds(epl), (size_t)SEL_BINS * 4);
        const int grid = kLearnBlocks + (plan.next_front ? h->sp_blocks : 0);
        switch (epl) {       // the front is that of step + 2: same parity as this step
            case 1: LAUNCH_ON(h, h->stream, lds, "tm_learn+sp_overlap", k_learn_overlap<1>, grid, RB, d, p, kLearnBlocks, bank, n_inputs, h->G, p, 2); break;
            case 2: LAUNCH_ON(h, h->stream, lds, "tm_learn+sp_overlap", k_learn_overlap<2>, grid, RB, d, p, kLearnBlocks, bank, n_inputs, h->G, p, 2); break;
            case 4: LAUNCH_ON(h, h->stream, lds, "tm_learn+sp_overlap", k_learn_overlap<4>, grid, RB, d, p, kLearnBlocks, bank, n_inputs, h->G, p, 2); break;
            default: LAUNCH_ON(h, h->stream, lds, "tm_learn+sp_overlap", k_learn_overlap<8>, grid, RB, d, p, kLearnBlocks, bank, n_inputs, h->G, p, 2); break;
        }
    }
    const int use_lds = scan_lds(d, 1) <= 64 * 1024;
    const int n_sel = plan.next_front ? 64 : 0, n_clear = plan.next_sp ? h->c256_blocks : 0;
    const size_t lds = std::max(scan_lds(d, use_lds), sizeof(SelShared));
    const int grid = h->scan_blocks + n_sel + n_clear;
    const int spec = scan_spec_blocks(h);
    if (scan_pool_is_large(h)) {
        if (use_lds) LAUNCH_ON(h, h->stream, lds, "tm_scan+sp_select", (k_scan_sel<true, 1>), grid, 256, d, p, n_sel, n_clear, p, spec);
        else LAUNCH_ON(h, h->stream, lds, "tm_scan+sp_select", (k_scan_sel<false, 1>), grid, 256, d, p, n_sel, n_clear, p, spec);
    } else {
        if (use_lds) LAUNCH_ON(h, h->stream, lds, "tm_scan+sp_select", (k_scan_sel<true, 6>), grid, 256, d, p, n_sel, n_clear, p, spec);
        else LAUNCH_ON(h, h->stream, lds, "tm_scan+sp_select", (k_scan_sel<false, 6>), grid, 256, d, p, n_sel, n_clear, p, spec);
    }
}

// work of a step that is not captured in its graph: the first step of a pipelined run has no SP work
// done for it; run the SP's step on its own, and the front of the next one
static void enqueue_cold_start(htm_handle *h, const uint32_t *bank, int n_inputs, int learning, StepPlan plan) {
    Dev &d = h->d;
    const int p = (int)(h->step_host & 1);
    if (plan.sp_done || !plan.next_sp) return;
    enqueue_sp_front(h, bank, n_inputs, p);
    enqueue_sp_back(h, bank, n_inputs, p, 1, EMIT_DUTY | EMIT_CLEAR, learning != 0);
    LAUNCH(h, "sp_overlap", k_sp_overlap, h->sp_blocks, RB, d, bank, n_inputs, h->G, p, p ^ 1, 1);
    LAUNCH(h, "sp_select", k_sel_pass, h->sel_blocks, RB, d, 1, p ^ 1);
}

static void enqueue_rest(htm_handle *h, int p, const uint32_t *bank, int n_inputs, int learning, StepPlan plan) {
    if (plan.sp_done || plan.next_sp) {
        enqueue_pipelined(h, p, learning, bank, n_inputs, plan);
    } else {                                        // one role per launch
        enqueue_sp_back(h, bank, n_inputs, p, 1, EMIT_ALL, false);
        enqueue_tm(h, h->d.k, learning, 1, p, bank, n_inputs, true);
    }
}

static int enqueue_step(htm_handle *h, const uint32_t *bank, int n_inputs, int learning, StepPlan plan) {
    if (!plan.sp_done && !plan.next_sp) enqueue_sp_front(h, bank, n_inputs, (int)(h->step_host & 1));
    enqueue_cold_start(h, bank, n_inputs, learning, plan);
    enqueue_rest(h, (int)(h->step_host & 1), bank, n_inputs, learning, plan);
    h->step_host += 1;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("kernel launch: ") + hipGetErrorString(e); return HTM_ERR_HIP; }
    return 0;
}

extern "C" int htm_abi_version(void) { return BITHTM_ABI_VERSION; }

extern "C" const char *htm_last_error(const htm_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" void htm_destroy(htm_handle *h) {
    if (!h) return;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    for (auto &kv : h->graphs) hipGraphExecDestroy(kv.second);
    for (auto &v : h->prof_events)
        for (auto &pr : v) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (hipEvent_t e : h->prof_all) hipEventDestroy(e);
    for (void *p : h->allocs) hipFree(p);
    if (h->seg_pinned) hipHostFree(h->seg_pinned);
    if (h->own_stream) hipStreamDestroy(h->stream);
    delete h;
}

static int fail_create(htm_handle *h, const std::string &msg, int code) {
    g_create_error = msg;
    if (h) { h->err = msg; htm_destroy(h); }
    return code;
}

extern "C" int htm_create(const htm_config *cfg, htm_handle **out) {
    if (!cfg || !out) return fail_create(nullptr, "htm_create: null argument", HTM_ERR_ARGUMENT);
    if (cfg->struct_bytes != sizeof(htm_config)) return fail_create(nullptr, "htm_create: htm_config size mismatch (ABI)", HTM_ERR_ARGUMENT);
    if (cfg->column_dim < 1 || cfg->active_columns < 1 || cfg->active_columns > cfg->column_dim)
        return fail_create(nullptr, "htm_create: need 1 <= active_columns <= column_dim", HTM_ERR_ARGUMENT);
    if (cfg->enable_sp && cfg->input_dim < 1) return fail_create(nullptr, "htm_create: input_dim < 1", HTM_ERR_ARGUMENT);
    if (cfg->enable_tm) {
        if (cfg->cell_dim < 1 || cfg->cell_dim > 32) return fail_create(nullptr, "htm_create: cell_dim must be in 1..32", HTM_ERR_ARGUMENT);
        if (cfg->segment_slots < 64 || cfg->segment_slots > MAX_SLOTS || cfg->segment_slots % 64)
            return fail_create(nullptr, "htm_create: segment_slots must be a multiple of 64 in 64..512", HTM_ERR_ARGUMENT);
        if (cfg->segment_capacity < 1) return fail_create(nullptr, "htm_create: segment_capacity < 1", HTM_ERR_ARGUMENT);
        if (cfg->segment_sampling_synapses < 1 || cfg->segment_sampling_synapses > 64)
            return fail_create(nullptr, "htm_create: segment_sampling_synapses must be in 1..64", HTM_ERR_ARGUMENT);
        if (cfg->segment_activation_threshold < cfg->segment_matching_threshold)      // projections.py:211
            return fail_create(nullptr, "htm_create: activation threshold < matching threshold", HTM_ERR_ARGUMENT);
        if ((long long)cfg->column_dim * 32 > 0x7FFFFFFFLL) return fail_create(nullptr, "htm_create: column_dim too large", HTM_ERR_ARGUMENT);
    }
    if (!cfg->enable_sp && !cfg->enable_tm) return fail_create(nullptr, "htm_create: nothing enabled", HTM_ERR_ARGUMENT);
    const int world = cfg->shard_world > 1 ? cfg->shard_world : 1;
    if (world > 1) {
        if (!cfg->enable_sp || !cfg->enable_tm) return fail_create(nullptr, "htm_create: a sharded handle needs SP and TM", HTM_ERR_ARGUMENT);
        if (cfg->shard_rank < 0 || cfg->shard_rank >= world) return fail_create(nullptr, "htm_create: shard_rank out of range", HTM_ERR_ARGUMENT);
        if (cfg->column_dim % (world * 64)) return fail_create(nullptr, "htm_create: column_dim must be a multiple of 64 * shard_world", HTM_ERR_ARGUMENT);
    }

    htm_handle *h = new htm_handle();
    h->cfg = *cfg;
    h->device = cfg->device;
    h->profile = false;
    h->prof_last = nullptr;
    h->step_host = 0;
    h->d_cols_stage = nullptr;
    h->rank = world > 1 ? cfg->shard_rank : 0;
    h->world = world;
    h->shard_bank = nullptr;
    h->shard_n_inputs = 1;
    h->shard_open = false;
    hipError_t e = hipSetDevice(cfg->device);
    if (e != hipSuccess) return fail_create(h, std::string("hipSetDevice: ") + hipGetErrorString(e), HTM_ERR_HIP);
    if (cfg->use_caller_stream) {
        h->stream = (hipStream_t)cfg->stream;          // NULL = the default stream
        h->own_stream = false;
    } else {
        e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e != hipSuccess) return fail_create(h, std::string("hipStreamCreate: ") + hipGetErrorString(e), HTM_ERR_HIP);
        h->own_stream = true;
    }
    Dev &d = h->d;
    memset(&d, 0, sizeof(d));
    d.I = cfg->enable_sp ? cfg->input_dim : 0;
    d.W = ((d.I + 127) / 128) * 4;
    d.W4 = d.W / 4;
    d.Ipad = d.W * 32;
    d.C = cfg->column_dim;
    d.world = world;
    d.c0 = h->rank * (d.C / world);
    d.c1 = d.c0 + d.C / world;
    d.K = cfg->enable_tm ? cfg->cell_dim : 0;
    d.k = cfg->active_columns;
    d.E = cfg->enable_tm ? cfg->segment_slots : 64;
    d.Scap = cfg->enable_tm ? cfg->segment_capacity : 0;
    d.work_cap = d.Scap + d.k * 32;
    d.sp_thr = cfg->sp_permanence_threshold; d.sp_don = cfg->sp_delta_on; d.sp_doff = cfg->sp_delta_off;
    d.coef = cfg->boost_coefficient; d.mom = cfg->duty_momentum; d.dinc = cfg->duty_increment;
    d.lrn_act = cfg->tm_learn_active; d.lrn_inact = cfg->tm_learn_inactive;
    d.pun_act = cfg->tm_punish_active; d.pun_inact = cfg->tm_punish_inactive;
    d.lrn_prune = cfg->tm_learn_prune; d.pun_prune = cfg->tm_punish_prune;
    d.perm_init = cfg->tm_permanence_initial; d.perm_thr = cfg->tm_permanence_threshold;
    d.act_thr = cfg->segment_activation_threshold; d.match_thr = cfg->segment_matching_threshold;
    d.sample = cfg->segment_sampling_synapses;
    d.seed = cfg->seed;

    int rc = 0;
    const size_t C = d.C, k = d.k;
    rc |= dalloc(h, &d.ctr, 1);
    rc |= dalloc(h, &d.active_cols[0], k + 8);     // (+8: the TM's list pass reads 8 entries per thread)
    rc |= dalloc(h, &d.active_cols[1], k + 8);
    if (cfg->enable_sp) {
        rc |= dalloc(h, &d.perm, C * d.Ipad);
        rc |= dalloc(h, &d.mask, C * d.W);
        rc |= dalloc(h, &d.duty, C);
        for (int q = 0; q < 2; ++q) {
            rc |= dalloc(h, &d.overlap[q], C);
            rc |= dalloc(h, &d.boosted[q], C);
            rc |= dalloc(h, &d.key[q], C);
        }
        rc |= dalloc(h, &d.hist, (size_t)2 * SEL_MAX_PASSES * SEL_BINS);
        rc |= dalloc(h, &d.hist0, (size_t)2 * HIST_REP * SEL_BINS);
        rc |= dalloc(h, &d.sel_blk, (C + 255) / 256);
        rc |= dalloc(h, &d.sel_rec, (C + 255) / 256 * 32);
        rc |= dalloc(h, &d.input_stage, (size_t)d.W);
    }
    if (cfg->enable_tm) {
        const size_t S = d.Scap, E = d.E;
        for (int q = 0; q < 2; ++q) {
            rc |= dalloc(h, &d.act[q], C);
            rc |= dalloc(h, &d.pred[q], C);
            rc |= dalloc(h, &d.winners[q], k * 32);
        }
        rc |= dalloc(h, &d.win[0], C);
        rc |= dalloc(h, &d.win[1], C);
        d.colwords = (int)((C + 63) / 64) * 2;
        rc |= dalloc(h, &d.colbits[0], (size_t)d.colwords);
        rc |= dalloc(h, &d.colbits[1], (size_t)d.colwords);
        rc |= dalloc(h, &d.bursting, k);
        rc |= dalloc(h, &d.winw_idx, k + 8);
        rc |= dalloc(h, &d.actcnt, k + 8);
        rc |= dalloc(h, &d.unacc_word, k + 8);
        rc |= dalloc(h, &d.unacc_list, k * 32);
        rc |= dalloc(h, &d.seg_cell, S);
        rc |= dalloc(h, &d.seg_nsyn, S);
        rc |= dalloc(h, &d.presyn, S * E);
        rc |= dalloc(h, &d.sperm, S * E);
        rc |= dalloc(h, &d.segcount, C * 32);
        rc |= dalloc(h, &d.cellmax, C * 32);
        rc |= dalloc(h, &d.seg_info, S);
        rc |= dalloc(h, &d.seg_jit, S);
        rc |= dalloc(h, &d.work, (size_t)d.work_cap);
        rc |= dalloc(h, &d.recyc_cnt, (S + 1023) / 1024);
        rc |= dalloc(h, &d.recyc_need, 2 * k * 32);
        rc |= dalloc(h, &d.dead_list, (size_t)1 + DEAD_CAP);
        if (world > 1) {
            rc |= dalloc(h, &d.spec_act, C);
            rc |= dalloc(h, &d.spec_win, C);
            rc |= dalloc(h, &d.spec_unacc, C);
            rc |= dalloc(h, &d.spec_burst, (C + 31) / 32);
        }
        rc |= dalloc(h, &h->d_cols_stage, k);
    }
    if (rc) return fail_create(h, h->err, HTM_ERR_HIP);
    // lanes per SP row: the smallest power of two >= W4, at most 64
    h->G = 1;
    h->seg_hint = 0;
    h->graph_steps = 16;
    if (const char *e = getenv("BITHTM_GRAPH_STEPS")) h->graph_steps = std::max(1, std::min(256, atoi(e)));
    h->seg_pinned = nullptr;
    if (hipHostMalloc((void **)&h->seg_pinned, sizeof(int), hipHostMallocDefault) == hipSuccess) *h->seg_pinned = 0; else h->seg_pinned = nullptr;
    while (h->G < d.W4 && h->G < 64) h->G <<= 1;
    // few fat blocks for the kernels that flush a histogram: every block adds into the same few
    // hot bins and same-address global atomics are slow (~88 per us per address)
    const int rows_per_block = (RB / 64) * 4 * (64 / h->G);   // waves x 4 row groups in flight
    h->sp_blocks = std::max(1, std::min((d.c1 - d.c0 + rows_per_block - 1) / rows_per_block, 256));
    d.trace = nullptr;
    if (getenv("BITHTM_TRACE")) rc |= dalloc(h, &d.trace, (size_t)8 * 4096 * 2);
    d.trace_until = getenv("BITHTM_TRACE_UNTIL") ? (uint32_t)strtoul(getenv("BITHTM_TRACE_UNTIL"), nullptr, 10) : 0xFFFFFFFFu;
    h->sel_blocks = std::max(1, std::min((d.C + RB - 1) / RB, 128));
    h->c256_blocks = (d.C + 255) / 256;
    h->s1024_blocks = std::max(1, (d.Scap + 1023) / 1024);
    h->scan_blocks = std::max(1, std::min((d.Scap + SCAN_SEGS - 1) / SCAN_SEGS, 2048));
    if (const char *e = getenv("BITHTM_SCAN_BLOCKS")) h->scan_blocks = std::max(1, atoi(e));      // tuning knob
    // boosted = float32 factor x integer overlap <= input_dim has at most 24 + bit_length(I)
    // significant bits, so the low 53 - 24 - bit_length(I) bits of every key are zero and the
    // radix passes that would only see them are skipped.
    {
        int B = 0;
        while ((1ll << B) <= (long long)d.I) ++B;
        const int informative = std::min(64, 64 - (29 - B));
        d.sel_passes = std::max(1, std::min(SEL_MAX_PASSES, (informative + SEL_DIGIT - 1) / SEL_DIGIT));
        d.low_zero = 64 - informative;            // key bits [0, low_zero) are zero in every key
        // grids of at most 1024 emit blocks finish the select inside k_sp_emit: two digits by launches
        if ((d.C + 255) / 256 <= 1024) d.sel_passes = std::min(d.sel_passes, 2);
        // test knobs: more launched digits (smaller buckets); fewer record slots (forces the fallback)
        if (const char *e = getenv("BITHTM_SEL_LAUNCH_DIGITS")) d.sel_passes = std::max(2, std::min(d.sel_passes, atoi(e)));
        d.cand_d = CAND_D;
        if (const char *e = getenv("BITHTM_CAND_D")) d.cand_d = std::max(0, std::min(CAND_D, atoi(e)));
        d.cand_pairwise = CAND_PAIRWISE;
        if (const char *e = getenv("BITHTM_CAND_PAIRWISE")) d.cand_pairwise = std::max(0, atoi(e));     // test knobs
        d.cand_others = CAND_OTHERS;
        if (const char *e = getenv("BITHTM_CAND_OTHERS")) d.cand_others = std::max(0, std::min(CAND_OTHERS, atoi(e)));
    }
    e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail_create(h, std::string("hipStreamSynchronize: ") + hipGetErrorString(e), HTM_ERR_HIP);
    *out = h;
    return HTM_OK;
}

extern "C" int htm_sync(htm_handle *h) {
    if (!h) return HTM_ERR_ARGUMENT;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return HTM_OK;
}

static int check_rows(htm_handle *h, const void *rows, int row_begin, int row_count) {
    if (!h) return HTM_ERR_ARGUMENT;
    if (!h->cfg.enable_sp) { h->err = "handle has no Spatial Pooler"; return HTM_ERR_STATE; }
    if (!rows || row_begin < 0 || row_count < 0 || row_begin + (long long)row_count > h->d.C) {
        h->err = "permanence rows out of range";
        return HTM_ERR_ARGUMENT;
    }
    return 0;
}

extern "C" int htm_sp_set_permanence(htm_handle *h, const double *rows, int32_t row_begin, int32_t row_count) {
    int rc = check_rows(h, rows, row_begin, row_count);
    if (rc) return rc;
    if (row_count == 0) return HTM_OK;
    Dev &d = h->d;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy2DAsync(d.perm + (size_t)row_begin * d.Ipad, (size_t)d.Ipad * 8, rows, (size_t)d.I * 8,
                               (size_t)d.I * 8, (size_t)row_count, hipMemcpyHostToDevice, h->stream));
    const long long waves = (long long)row_count * (d.Ipad / 64);
    const int blocks = (int)std::min<long long>((waves + 3) / 4, 8192);
    hipLaunchKernelGGL(k_sp_build_mask, dim3(blocks), dim3(256), 0, h->stream, d, row_begin, row_count);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return HTM_OK;
}

extern "C" int htm_sp_get_permanence(htm_handle *h, double *rows, int32_t row_begin, int32_t row_count) {
    int rc = check_rows(h, rows, row_begin, row_count);
    if (rc) return rc;
    if (row_count == 0) return HTM_OK;
    Dev &d = h->d;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy2DAsync(rows, (size_t)d.I * 8, d.perm + (size_t)row_begin * d.Ipad, (size_t)d.Ipad * 8,
                               (size_t)d.I * 8, (size_t)row_count, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return HTM_OK;
}

static int stage_input(htm_handle *h, const uint32_t *packed_input) {
    Dev &d = h->d;
    const int words = (d.I + 31) / 32;
    HIPCHK(h, hipMemcpyAsync(d.input_stage, packed_input, (size_t)words * 4, hipMemcpyHostToDevice, h->stream));
    return 0;
}

extern "C" int htm_step(htm_handle *h, const uint32_t *packed_input, int32_t learning) {
    if (!h || !packed_input) return HTM_ERR_ARGUMENT;
    if (!h->cfg.enable_sp || !h->cfg.enable_tm) { h->err = "htm_step needs a handle with SP and TM"; return HTM_ERR_STATE; }
    if (h->world > 1) { h->err = "sharded handle: use htm_shard_begin / htm_shard_finish"; return HTM_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    int rc = stage_input(h, packed_input);
    if (rc) return rc;
    return enqueue_step(h, h->d.input_stage, 1, learning ? 1 : 0, StepPlan{false, false, false});
}

extern "C" int htm_sp_step(htm_handle *h, const uint32_t *packed_input, int32_t learning) {
    if (!h || !packed_input) return HTM_ERR_ARGUMENT;
    if (!h->cfg.enable_sp) { h->err = "handle has no Spatial Pooler"; return HTM_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    int rc = stage_input(h, packed_input);
    if (rc) return rc;
    const int p = (int)(h->step_host & 1);
    enqueue_sp_front(h, h->d.input_stage, 1, p);
    enqueue_sp_back(h, h->d.input_stage, 1, p, 0, EMIT_ALL, learning && !h->cfg.enable_tm);
    h->step_host += 1;
    return HTM_OK;
}

extern "C" int htm_tm_step(htm_handle *h, const int32_t *active_column, int32_t n, int32_t learning, int32_t return_winner_cell) {
    if (!h || (!active_column && n > 0)) return HTM_ERR_ARGUMENT;
    if (!h->cfg.enable_tm) { h->err = "handle has no Temporal Memory"; return HTM_ERR_STATE; }
    Dev &d = h->d;
    if (n < 0 || n > d.k) { h->err = "htm_tm_step: more active columns than active_columns"; return HTM_ERR_ARGUMENT; }
    std::vector<int> cols(active_column, active_column + n);
    std::sort(cols.begin(), cols.end());
    for (int i = 0; i < n; ++i)
        if (cols[i] < 0 || cols[i] >= d.C || (i && cols[i] == cols[i - 1])) { h->err = "htm_tm_step: bad active column list"; return HTM_ERR_ARGUMENT; }
    HIPCHK(h, hipSetDevice(h->device));
    if (n) HIPCHK(h, hipMemcpyAsync(h->d_cols_stage, cols.data(), (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));      // cols is a local
    const int p = (int)(h->step_host & 1);
    const int want = (learning || return_winner_cell) ? 1 : 0;
    LAUNCH(h, "tm_load_active", k_tm_load_active, std::min((d.C + 255) / 256, 1024), 256, d, p, h->d_cols_stage, n);
    LAUNCH(h, "tm_activate", k_tm_activate, std::max(1, (n * 32 + 255) / 256), 256, d, p, n, want);
    enqueue_tm(h, n, learning ? 1 : 0, want, p, nullptr, 1, false);
    h->step_host += 1;
    return HTM_OK;
}

extern "C" int htm_run(htm_handle *h, const uint32_t *device_inputs, int32_t n_inputs, int32_t n_steps, int32_t learning, int32_t use_graph) {
    if (!h || !device_inputs || n_inputs < 1 || n_steps < 0) return HTM_ERR_ARGUMENT;
    if (!h->cfg.enable_sp || !h->cfg.enable_tm) { h->err = "htm_run needs a handle with SP and TM"; return HTM_ERR_STATE; }
    if (h->world > 1) { h->err = "sharded handle: use htm_shard_begin / htm_shard_finish"; return HTM_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    learning = learning ? 1 : 0;
    const bool graph = (use_graph & 1) && !h->profile;
    const bool pipeline = !(use_graph & 2) && can_pipeline(h);
    // Graphs hold the launches of one step, or of kGraphSteps consecutive steady-state steps (a graph
    // launch boundary costs about 5 us more than a kernel boundary inside a graph: tools/step_timeline.py).
    // Nothing in a graph depends on the step index: kernels read it, and with it the bank row, from
    // the device counter.
    const int kGraphSteps = h->graph_steps;
    if (h->seg_pinned) { const int seen = *(volatile int *)h->seg_pinned; h->seg_hint = std::max(h->seg_hint, seen); }      // what the last run left
    bool sp_done = false;                           // the SP has already done the coming step
    for (int t = 0; t < n_steps;) {
        const StepPlan plan{sp_done, pipeline && t + 1 < n_steps, pipeline && t + 2 < n_steps};
        sp_done = plan.next_sp;
        if (!graph) {
            int rc = enqueue_step(h, device_inputs, n_inputs, learning, plan);
            if (rc) return rc;
            t += 1;
            continue;
        }
        const int p = (int)(h->step_host & 1);
        // steady state: this and the next kGraphSteps - 1 steps all look ahead fully
        const int span = (plan.sp_done && plan.next_front && t + kGraphSteps + 1 < n_steps) ? kGraphSteps : 1;
        if (!plan.sp_done && !plan.next_sp) enqueue_sp_front(h, device_inputs, n_inputs, p);    // eager
        enqueue_cold_start(h, device_inputs, n_inputs, learning, plan);                         // eager: first step of a pipelined run
        auto key = std::make_tuple(p, learning * 16 + (span > 1 ? 8 : 0) + (plan.sp_done ? 4 : 0) + (plan.next_sp ? 2 : 0) + (plan.next_front ? 1 : 0) + 32 * scan_spec_blocks(h) + (scan_pool_is_large(h) ? (1 << 20) : 0),
                                   (const void *)device_inputs, n_inputs);
        auto it = h->graphs.find(key);
        if (it == h->graphs.end()) {
            hipGraph_t graph_obj;
            HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < span; ++i) enqueue_rest(h, (p + i) & 1, device_inputs, n_inputs, learning, plan);
            hipError_t e = hipStreamEndCapture(h->stream, &graph_obj);
            if (e != hipSuccess) { h->err = std::string("hipStreamEndCapture: ") + hipGetErrorString(e); return HTM_ERR_HIP; }
            hipGraphExec_t exec;
            HIPCHK(h, hipGraphInstantiate(&exec, graph_obj, nullptr, nullptr, 0));
            hipGraphDestroy(graph_obj);
            it = h->graphs.emplace(key, exec).first;
        }
        HIPCHK(h, hipGraphLaunch(it->second, h->stream));
        h->step_host += span;
        t += span;
    }
    // leave the segment count where the next call finds it (no wait: it may see the one before)
    if (h->seg_pinned && n_steps > 0) HIPCHK(h, hipMemcpyAsync(h->seg_pinned, &h->d.ctr->S, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    return HTM_OK;
}

extern "C" int64_t htm_shard_record_bytes(htm_handle *h) {
    if (!h) return HTM_ERR_ARGUMENT;
    return (int64_t)shard_record_bytes(h->d.c1 - h->d.c0);
}

extern "C" int htm_shard_begin(htm_handle *h, const uint32_t *device_inputs, int32_t n_inputs, const uint32_t *packed_input,
                               int32_t learning, void *send_device) {
    if (!h || !send_device || (!device_inputs == !packed_input)) return HTM_ERR_ARGUMENT;
    if (h->world < 2) { h->err = "htm_shard_begin: handle is not sharded"; return HTM_ERR_STATE; }
    if (h->shard_open) { h->err = "htm_shard_begin: previous step not finished"; return HTM_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    if (packed_input) {
        int rc = stage_input(h, packed_input);
        if (rc) return rc;
        h->shard_bank = h->d.input_stage;
        h->shard_n_inputs = 1;
    } else {
        if (n_inputs < 1) return HTM_ERR_ARGUMENT;
        h->shard_bank = device_inputs;
        h->shard_n_inputs = n_inputs;
    }
    Dev &d = h->d;
    const int p = (int)(h->step_host & 1);
    const int cl = d.c1 - d.c0;
    LAUNCH(h, "sp_overlap", k_sp_overlap, h->sp_blocks, RB, d, h->shard_bank, h->shard_n_inputs, h->G, p, p, 0);
    LAUNCH(h, "shard_pack_clear", k_shard_pack_clear, 1, 256, d, (unsigned char *)send_device);
    LAUNCH(h, "shard_pack", k_shard_pack, (cl * 32 + 255) / 256, 256, d, p, (unsigned char *)send_device);
    h->shard_open = true;
    (void)learning;
    return HTM_OK;
}

extern "C" int htm_shard_finish(htm_handle *h, const void *recv_device, int32_t learning) {
    if (!h || !recv_device) return HTM_ERR_ARGUMENT;
    if (h->world < 2 || !h->shard_open) { h->err = "htm_shard_finish: no step in progress"; return HTM_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    Dev &d = h->d;
    const int p = (int)(h->step_host & 1);
    learning = learning ? 1 : 0;
    LAUNCH(h, "shard_unpack", k_shard_unpack, h->sel_blocks, 1024, d, (const unsigned char *)recv_device, h->rank, p);
    for (int pass = 1; pass < d.sel_passes; ++pass) LAUNCH(h, "sp_select", k_sel_pass, h->sel_blocks, RB, d, pass, p);
    const int fused = h->c256_blocks <= 1024;
    if (!fused) LAUNCH(h, "sp_count", k_sp_count, h->c256_blocks, 256, d, p);
    LAUNCH(h, "sp_emit", k_sp_emit, h->c256_blocks, 256, d, p, 1, fused, EMIT_ALL);
    enqueue_tm(h, d.k, learning, 1, p, h->shard_bank, h->shard_n_inputs, true);
    h->step_host += 1;
    h->shard_open = false;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("kernel launch: ") + hipGetErrorString(e); return HTM_ERR_HIP; }
    return HTM_OK;
}

extern "C" int htm_bank_upload(htm_handle *h, const uint32_t *host_inputs, int32_t n_inputs, uint32_t **device_bank) {
    if (!h || !host_inputs || n_inputs < 1 || !device_bank) return HTM_ERR_ARGUMENT;
    if (!h->cfg.enable_sp) { h->err = "handle has no Spatial Pooler"; return HTM_ERR_STATE; }
    Dev &d = h->d;
    HIPCHK(h, hipSetDevice(h->device));
    uint32_t *bank = nullptr;
    int rc = dalloc(h, &bank, (size_t)n_inputs * d.W);
    if (rc) return rc;
    const size_t words = (size_t)(d.I + 31) / 32;
    HIPCHK(h, hipMemcpy2DAsync(bank, (size_t)d.W * 4, host_inputs, words * 4, words * 4, (size_t)n_inputs,
                               hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *device_bank = bank;
    return HTM_OK;
}

static int read_counters(htm_handle *h, Counters *out) {
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, h->d.ctr, sizeof(Counters), hipMemcpyDeviceToHost));
    h->seg_hint = out->S;                           // exact: the stream is idle
    if (h->seg_pinned) *h->seg_pinned = out->S;
    return 0;
}

extern "C" int htm_get_info(htm_handle *h, htm_info *out) {
    if (!h || !out) return HTM_ERR_ARGUMENT;
    Counters c;
    int rc = read_counters(h, &c);
    if (rc) return rc;
    const int q = (int)((h->step_host + 1) & 1);          // parity of the last completed step
    out->step_index = h->step_host;
    out->segments = c.S;
    out->matching_segments = 0;
    if (c.has_distal && c.S > 0) {
        std::vector<uint32_t> info((size_t)c.S);
        HIPCHK(h, hipMemcpy(info.data(), h->d.seg_info, info.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t v : info) out->matching_segments += (v >> 30) & 1u;
    }
    out->winner_cells = c.n_win[q];
    out->active_cells = c.n_active_cells;
    out->has_distal_state = c.has_distal;
    out->has_winner_cells = h->step_host > 0 ? c.has_winner[q] : 0;
    out->capacity_error = c.error;
    out->words_per_row = h->d.W;
    out->new_segment_requests = c.n_un;
    out->recycled_segments = c.n_un ? c.n_recycled : 0;
    out->appended_segments = c.n_un ? c.n_new : 0;
    out->work_items = c.n_work_last;
    out->select_fallbacks = c.sel_fallbacks;
    if (c.error) {
        h->err = std::string("capacity exhausted:") + ((c.error & 1) ? " segment pool (segment_capacity)" : "") +
                 ((c.error & 2) ? " synapse slots (segment_slots)" : "") + ((c.error & 4) ? " work list / growth staging" : "") +
                 ((c.error & 8) ? " dead-segment report (DEAD_CAP)" : "") +
                 ((c.error & 16) ? " (internal) block hand-off timed out in k_sp_emit" : "");
        return HTM_ERR_CAPACITY;          // *out is filled in all the same
    }
    return HTM_OK;
}

// conversions between the internal cell encoding (col*32+cell) and the ABI's flat ids
static inline int enc_flat(int enc, int K) { return (enc >> 5) * K + (enc & 31); }
static inline int flat_enc(int flat, int K) { return (flat / K) * 32 + (flat % K); }

extern "C" int64_t htm_read(htm_handle *h, int32_t field, void *dst, int64_t count) {
    if (!h || !dst) return HTM_ERR_ARGUMENT;
    Counters c;
    int rc = read_counters(h, &c);
    if (rc) return rc;
    Dev &d = h->d;
    const int q = (int)((h->step_host + 1) & 1);
    const bool sp = h->cfg.enable_sp, tm = h->cfg.enable_tm;
    const int64_t C = d.C, K = d.K, S = c.S, E = d.E;
    auto need = [&](bool ok, int64_t n) -> int64_t {
        if (!ok) { h->err = "htm_read: field not available on this handle"; return HTM_ERR_STATE; }
        if (count < n) { h->err = "htm_read: buffer too small"; return HTM_ERR_ARGUMENT; }
        return n;
    };
    auto copy = [&](const void *src, int64_t n, size_t elem) -> int64_t {
        if (n > 0 && hipMemcpy(dst, src, (size_t)n * elem, hipMemcpyDeviceToHost) != hipSuccess) { h->err = "htm_read: hipMemcpy failed"; return HTM_ERR_HIP; }
        return n;
    };
    int64_t n;
    switch (field) {
        case HTM_F_ACTIVE_COLUMN: if ((n = need(true, d.k)) < 0) return n; return copy(d.active_cols[q], n, 4);
        case HTM_F_OVERLAPS: if ((n = need(sp, C)) < 0) return n; return copy(d.overlap[q], n, 4);
        case HTM_F_BOOSTED: if ((n = need(sp, C)) < 0) return n; return copy(d.boosted[q], n, 8);
        case HTM_F_DUTY_CYCLE: if ((n = need(sp, C)) < 0) return n; return copy(d.duty, n, 4);
        case HTM_F_CELL_ACTIVATION: if ((n = need(tm, C)) < 0) return n; return copy(d.act[q], n, 4);
        case HTM_F_CELL_PREDICTION: if ((n = need(tm, C)) < 0) return n; return copy(d.pred[q], n, 4);
        case HTM_F_WINNER_WORDS: if ((n = need(tm, C)) < 0) return n; return copy(d.win[q], n, 4);
        case HTM_F_BURSTING: if ((n = need(tm, d.k)) < 0) return n; return copy(d.bursting, n, 1);
        case HTM_F_SEG_NSYN: if ((n = need(tm, S)) < 0) return n; return copy(d.seg_nsyn, n, 4);
        case HTM_F_SEG_POTENTIAL:
        case HTM_F_MATCH_SEGMENT:
        case HTM_F_MATCH_INFO:
        case HTM_F_MATCH_JITTER: {
            // the device keeps one info word per segment; the matching-segment lists of
            // PredictiveProjection.State (ascending ids, projections.py:247) are its non-zero part
            if (!tm) { h->err = "htm_read: field not available on this handle"; return HTM_ERR_STATE; }
            std::vector<uint32_t> info((size_t)S);
            std::vector<float> jit((size_t)S);
            if (S && (hipMemcpy(info.data(), d.seg_info, (size_t)S * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                      hipMemcpy(jit.data(), d.seg_jit, (size_t)S * 4, hipMemcpyDeviceToHost) != hipSuccess)) { h->err = "htm_read: hipMemcpy failed"; return HTM_ERR_HIP; }
            if (!c.has_distal) std::fill(info.begin(), info.end(), 0u);
            int64_t m = 0;
            if (field == HTM_F_SEG_POTENTIAL) {
                if ((n = need(true, S)) < 0) return n;
                for (int64_t i = 0; i < S; ++i) ((int *)dst)[i] = (int)(info[(size_t)i] & 0xFFFu);
                return S;
            }
            for (int64_t i = 0; i < S; ++i) m += (info[(size_t)i] >> 30) & 1u;
            if ((n = need(true, m)) < 0) return n;
            m = 0;
            for (int64_t i = 0; i < S; ++i) {
                const uint32_t v = info[(size_t)i];
                if (!((v >> 30) & 1u)) continue;
                if (field == HTM_F_MATCH_SEGMENT) ((int *)dst)[m] = (int)i;
                else if (field == HTM_F_MATCH_INFO) ((uint32_t *)dst)[m] = v & ~0x40000000u;
                else ((float *)dst)[m] = jit[(size_t)i];
                ++m;
            }
            return m;
        }
        case HTM_F_WINNER_CELL:
        case HTM_F_SEG_CELL: {
            const bool w = field == HTM_F_WINNER_CELL;
            if ((n = need(tm, w ? c.n_win[q] : S)) < 0) return n;
            if ((rc = (int)copy(w ? d.winners[q] : d.seg_cell, n, 4)) < 0) return rc;
            int *v = (int *)dst;
            for (int64_t i = 0; i < n; ++i) v[i] = enc_flat(v[i], (int)K);
            return n;
        }
        case HTM_F_SEG_PRESYN:
        case HTM_F_SEG_PERM: {
            if ((n = need(tm, S * E)) < 0) return n;
            std::vector<int> nsyn((size_t)S);
            if (S && hipMemcpy(nsyn.data(), d.seg_nsyn, (size_t)S * 4, hipMemcpyDeviceToHost) != hipSuccess) { h->err = "htm_read: hipMemcpy failed"; return HTM_ERR_HIP; }
            if ((rc = (int)copy(field == HTM_F_SEG_PRESYN ? (const void *)d.presyn : (const void *)d.sperm, n, 4)) < 0) return rc;
            for (int64_t s = 0; s < S; ++s)
                for (int64_t e = 0; e < E; ++e) {
                    const bool valid = e < nsyn[(size_t)s];
                    if (field == HTM_F_SEG_PRESYN) { int *v = (int *)dst + s * E + e; *v = valid ? enc_flat(*v, (int)K) : -1; }
                    else if (!valid) ((float *)dst)[s * E + e] = -1.0f;
                }
            return n;
        }
        case HTM_F_SEGCOUNT:
        case HTM_F_CELL_MAX_JITTER: {
            if ((n = need(tm, C * K)) < 0) return n;
            std::vector<uint32_t> tmp((size_t)C * 32);
            const void *src = field == HTM_F_SEGCOUNT ? (const void *)d.segcount : (const void *)d.cellmax;
            if (hipMemcpy(tmp.data(), src, tmp.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { h->err = "htm_read: hipMemcpy failed"; return HTM_ERR_HIP; }
            uint32_t *v = (uint32_t *)dst;
            for (int64_t col = 0; col < C; ++col)
                for (int64_t j = 0; j < K; ++j) v[col * K + j] = tmp[(size_t)col * 32 + j];
            return n;
        }
        default: h->err = "htm_read: unknown field"; return HTM_ERR_ARGUMENT;
    }
}

extern "C" int htm_write(htm_handle *h, int32_t field, const void *src, int64_t count) {
    if (!h || (!src && count > 0) || count < 0) return HTM_ERR_ARGUMENT;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    Dev &d = h->d;
    const bool sp = h->cfg.enable_sp, tm = h->cfg.enable_tm;
    const int64_t C = d.C, K = d.K, E = d.E;
    const int q = (int)((h->step_host + 1) & 1);      // becomes "previous step" for the next one
    auto put = [&](void *dstp, const void *s, int64_t n, size_t elem, int64_t cap) -> int {
        if (n > cap) { h->err = "htm_write: too many elements"; return HTM_ERR_ARGUMENT; }
        if (n > 0 && hipMemcpy(dstp, s, (size_t)n * elem, hipMemcpyHostToDevice) != hipSuccess) { h->err = "htm_write: hipMemcpy failed"; return HTM_ERR_HIP; }
        return HTM_OK;
    };
    if (((field >= HTM_F_CELL_ACTIVATION) && !tm) || ((field >= HTM_F_OVERLAPS && field <= HTM_F_DUTY_CYCLE) && !sp)) {
        h->err = "htm_write: field not available on this handle";
        return HTM_ERR_STATE;
    }
    switch (field) {
        case HTM_F_DUTY_CYCLE: return put(d.duty, src, count, 4, C);
        case HTM_F_CELL_ACTIVATION: return put(d.act[q], src, count, 4, C);
        case HTM_F_CELL_PREDICTION: return put(d.pred[q], src, count, 4, C);
        case HTM_F_SEG_NSYN: return put(d.seg_nsyn, src, count, 4, d.Scap);
        case HTM_F_SEG_POTENTIAL: h->imp_pot.assign((const int *)src, (const int *)src + count); return HTM_OK;
        case HTM_F_MATCH_SEGMENT: h->imp_match_seg.assign((const int *)src, (const int *)src + count); return HTM_OK;
        case HTM_F_MATCH_INFO: h->imp_match_info.assign((const uint32_t *)src, (const uint32_t *)src + count); return HTM_OK;
        case HTM_F_MATCH_JITTER: h->imp_match_jit.assign((const float *)src, (const float *)src + count); return HTM_OK;
        case HTM_F_SEG_PERM: return put(d.sperm, src, count, 4, (int64_t)d.Scap * E);
        case HTM_F_WINNER_CELL:
        case HTM_F_SEG_CELL:
        case HTM_F_SEG_PRESYN: {
            std::vector<int> v((const int *)src, (const int *)src + count);
            for (auto &x : v) x = x < 0 ? 0 : flat_enc(x, (int)K);
            if (field == HTM_F_WINNER_CELL) return put(d.winners[q], v.data(), count, 4, (int64_t)d.k * 32);
            if (field == HTM_F_SEG_CELL) return put(d.seg_cell, v.data(), count, 4, d.Scap);
            return put(d.presyn, v.data(), count, 4, (int64_t)d.Scap * E);
        }
        case HTM_F_SEGCOUNT:
        case HTM_F_CELL_MAX_JITTER: {
            if (count != C * K) { h->err = "htm_write: need column_dim * cell_dim elements"; return HTM_ERR_ARGUMENT; }
            std::vector<uint32_t> tmp((size_t)C * 32, 0u);
            const uint32_t *v = (const uint32_t *)src;
            for (int64_t col = 0; col < C; ++col)
                for (int64_t j = 0; j < K; ++j) tmp[(size_t)col * 32 + j] = v[col * K + j];
            return put(field == HTM_F_SEGCOUNT ? (void *)d.segcount : (void *)d.cellmax, tmp.data(), (int64_t)tmp.size(), 4, (int64_t)tmp.size());
        }
        default: h->err = "htm_write: field is not writable"; return HTM_ERR_ARGUMENT;
    }
}

extern "C" int htm_import_begin(htm_handle *h, int64_t step_index) {
    if (!h || step_index < 0) return HTM_ERR_ARGUMENT;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->step_host = step_index;
    return HTM_OK;
}

extern "C" int htm_import_commit(htm_handle *h, int32_t segments, int32_t matching_segments, int32_t winner_cells,
                                 int32_t has_distal_state, int32_t has_winner_cells) {
    if (!h) return HTM_ERR_ARGUMENT;
    Dev &d = h->d;
    if (segments < 0 || segments > d.Scap || matching_segments < 0 || matching_segments > segments ||
        winner_cells < 0 || winner_cells > d.k * 32) { h->err = "htm_import_commit: bad scalars"; return HTM_ERR_ARGUMENT; }
    Counters c;
    int rc = read_counters(h, &c);
    if (rc) return rc;
    const int q = (int)((h->step_host + 1) & 1);
    c.step[h->step_host & 1] = (uint32_t)h->step_host;
    c.n_work = 0;
    c.n_bind = 0;
    c.S = segments;
    h->seg_hint = segments;                         // (the one place where the count can go down)
    if (h->seg_pinned) *h->seg_pinned = segments;
    {   // dense per-segment info from the staged PredictiveProjection.State lists
        const size_t M = (size_t)matching_segments;
        if (has_distal_state && (h->imp_pot.size() != (size_t)segments || h->imp_match_seg.size() != M ||
                                 h->imp_match_info.size() != M || h->imp_match_jit.size() != M)) {
            h->err = "htm_import_commit: SEG_POTENTIAL / MATCH_* fields missing or of the wrong length";
            return HTM_ERR_ARGUMENT;
        }
        std::vector<uint32_t> info((size_t)segments, 0u);
        std::vector<float> jit((size_t)segments, 0.f);
        if (has_distal_state) {
            for (size_t i = 0; i < (size_t)segments; ++i) info[i] = (uint32_t)h->imp_pot[i] & 0xFFFu;
            for (size_t i = 0; i < M; ++i) {
                const int sgm = h->imp_match_seg[i];
                if (sgm < 0 || sgm >= segments) { h->err = "htm_import_commit: matching segment id out of range"; return HTM_ERR_ARGUMENT; }
                info[(size_t)sgm] = (h->imp_match_info[i] & ~0x40000000u) | 0x40000000u;
                jit[(size_t)sgm] = h->imp_match_jit[i];
            }
        }
        if (segments) {
            HIPCHK(h, hipMemcpy(d.seg_info, info.data(), info.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(d.seg_jit, jit.data(), jit.size() * 4, hipMemcpyHostToDevice));
        }
        h->imp_pot.clear(); h->imp_match_seg.clear(); h->imp_match_info.clear(); h->imp_match_jit.clear();
    }
    c.n_win[q] = winner_cells;
    c.has_winner[q] = has_winner_cells ? 1 : 0;
    c.has_distal = has_distal_state ? 1 : 0;
    c.error = 0;
    HIPCHK(h, hipMemcpy(d.ctr, &c, sizeof(c), hipMemcpyHostToDevice));
    if (h->cfg.enable_tm) {
        hipLaunchKernelGGL(k_tm_recount, dim3(h->s1024_blocks), dim3(256), 0, h->stream, d);
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return HTM_OK;
}

extern "C" int htm_profile(htm_handle *h, int32_t enable) {
    if (!h) return HTM_ERR_ARGUMENT;
    h->profile = enable != 0;
    return HTM_OK;
}

extern "C" int64_t htm_trace_read(htm_handle *h, uint64_t *dst, int64_t count) {
    if (!h || !dst) return HTM_ERR_ARGUMENT;
    if (!h->d.trace) { h->err = "htm_trace_read: handle was not created with BITHTM_TRACE=1"; return HTM_ERR_STATE; }
    const int64_t n = (int64_t)8 * 4096 * 2;
    if (count < n) { h->err = "htm_trace_read: buffer too small"; return HTM_ERR_ARGUMENT; }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(dst, h->d.trace, (size_t)n * 8, hipMemcpyDeviceToHost));
    return n;
}

extern "C" int htm_profile_read(htm_handle *h, int32_t max_kernels, const char **names, double *total_ms, int64_t *launches) {
    if (!h) return HTM_ERR_ARGUMENT;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < h->prof_names.size(); ++i) {
        for (auto &pr : h->prof_events[i]) {
            float ms = 0.f;
            hipEventElapsedTime(&ms, pr.first, pr.second);
            h->prof_ms[i] += ms;
            h->prof_n[i] += 1;
            h->prof_all.push_back(pr.first);
            h->prof_all.push_back(pr.second);
        }
        h->prof_events[i].clear();
    }
    for (hipEvent_t e : h->prof_all) hipEventDestroy(e);
    h->prof_all.clear();
    int n = (int)std::min<size_t>(h->prof_names.size(), (size_t)std::max(max_kernels, 0));
    for (int i = 0; i < n; ++i) {
        if (names) names[i] = h->prof_names[i].c_str();
        if (total_ms) total_ms[i] = h->prof_ms[i];
        if (launches) launches[i] = h->prof_n[i];
    }
    for (size_t i = 0; i < h->prof_names.size(); ++i) { h->prof_ms[i] = 0; h->prof_n[i] = 0; }
    return n;
}
