// Spatial Pooler roles and kernels: mask build, overlap + boost, radix select, select finish + emit (with the
// Temporal Memory per-column activation it ends in), permanence rows; the two kernels around the sharded exchange.
// Part of the single translation unit htm_engine.hip (included there, in this order:
// htm_dev.h, htm_sp_kernels.h, htm_tm_kernels.h, htm_pipeline.h).
#ifndef BITHTM_HTM_SP_KERNELS_H
#define BITHTM_HTM_SP_KERNELS_H

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
// Sharded handles run it on their own rows only: the select that follows picks their own candidates.
// Roles are written against (blk, nblk) instead of blockIdx / gridDim so that two independent
// roles can share one launch (pipelined schedule: step t's TM work beside step t+1's SP work).
// sp = parity buffer of the SP step being computed; step_offset = that step minus the current one.
// diagnostic build (-DBITHTM_OVERLAP_STAMPS, handle created under BITHTM_TRACE=1): device clock at the phases of every
// overlap block of the three-launch schedule, d.trace[3 * 8192 + block * 8 + phase] (tools/overlap_phases.py)
#ifdef BITHTM_OVERLAP_STAMPS
#define OV_STAMP(i) do { if (d.trace && wmode && threadIdx.x == 0 && blk < 512) d.trace[(size_t)3 * 8192 + (size_t)blk * 8 + (i)] = wall_clock64(); } while (0)
#else
#define OV_STAMP(i) do { } while (0)
#endif
template <int BS>
// wmode: the histogram of the windowed select (win_bin) instead of the top key digit
// fold (two-launch schedule, k_act_mid_rows: the overlap of step p's successor in the launch that updates step p's permanence
// rows): the role also applies step p's duty-cycle update (regularizations.py:19-21) to every column it finishes -- 1: all of
// them; 2: not the winners of step p (colbits[p]), whose rows are being rewritten in this very launch: their row blocks count
// the new row against the coming input themselves (role_sp_row, OWN) and leave key, bin and duty cycle
__device__ __forceinline__ void role_overlap(const Dev &d, const uint32_t *__restrict__ bank, int n_inputs, int G,
                                             int p, int sp, int step_offset, int blk, int nblk, uint32_t *h, int wmode = 0, int fold = 0) {
    const int gtid = blk * BS + threadIdx.x;
    const int nthreads = nblk * BS;
    uint32_t *ghist = d.hist + sp * SEL_MAX_PASSES * SEL_BINS;
    OV_STAMP(0);
    if (gtid == 0) d.ctr->emit_epoch += 1;          // a new generation of k_sp_emit records
    // (a bank of one row -- the host-fed step's input, the row the three-launch schedule stages ahead -- is read where it is:
    // its loads do not wait for the step counter)
    const uint4 *in4 = (const uint4 *)bank;
    if (n_inputs > 1) in4 = (const uint4 *)(bank + (size_t)((d.ctr->step[p] + (uint32_t)step_offset) % (uint32_t)n_inputs) * d.W);
    const uint4 *mask4 = (const uint4 *)d.mask;
    const int lane = lane_id();
    const int rpw = 64 / G, sub = lane / G, l = lane % G;
    const int wave = gtid >> 6, nwaves = nthreads >> 6;
    constexpr int U = 4;                           // row groups in flight per wave
    // With 8 lanes to a row (inputs of up to 1 024 bits: G == 8) a wave's 32 rows are finished by 32 of its lanes, one row
    // each -- lane l < 4 of group `sub` takes row group u = l: the group's sum is in all of its lanes --, not by the rows'
    // first lanes four times over (the exponential, the key and its bin were a microsecond of a wave's instructions that way).
    const bool spread = G == 8;
    const int first_row0 = d.c0 + wave * rpw * U;
    // the first rows' masks and duty cycles are asked for before anything else: the LDS histogram is zeroed while they travel
    uint4 m[U];                                    // (the first row group's first chunk; the loop below loads every other into it)
    float dty_first = 0.f;                         // (a lane that finishes ONE row -- spread -- asks for its duty cycle here; the
                                                   // first lane of a wider group asks for its four when the counts are in)
    uint32_t cbw_first = 0u;                       // (fold: the word of step p's column bitmap that holds the row's bit -- asked for
                                                   // with the duty cycle where a lane finishes ONE row; else when the row is finished)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int row = first_row0 + u * rpw + sub;
        m[u] = (row < d.c1 && l < d.W4) ? mask4[(uint32_t)row * (uint32_t)d.W4 + (uint32_t)l] : make_uint4(0, 0, 0, 0);
    }
    if (spread) {
        const int row = first_row0 + (l & 3) * rpw + sub;
        dty_first = (l < U && row < d.c1) ? d.duty[row] : 0.f;
        if (fold) cbw_first = (l < U && row < d.c1) ? d.colbits[p][row >> 5] : 0u;
    }
    for (int i = gtid; i < (d.sel_passes - 1) * SEL_BINS; i += nthreads) ghist[SEL_BINS + i] = 0;
    for (int i = threadIdx.x; i < SEL_BINS; i += BS) h[i] = 0;
    if (gtid == 0) {
        d.ctr->sel_pass_prefix[sp][0] = 0;
        d.ctr->sel_pass_krem[sp][0] = (uint32_t)d.sel_k;
    }
    const uint32_t wbase = wmode ? d.ctr->sel_win[sp] : 0u;
    lds_barrier();                                 // (LDS only: the zeroing stores above are for later launches, the loads stay in flight)
    OV_STAMP(1);
    static_assert(WIN_BINS <= SEL_BINS - SEL_COARSE, "the runs' sums live behind the window's bins in the LDS histogram");
    uint32_t *hc = h + SEL_BINS - SEL_COARSE;       // (window mode only)
    // one row's share of the step: overlap (projections.py:18-21), boosted overlap (regularizations.py:15-17), key, bin
    auto finish_row = [&](bool owner, int row, int cn, float duty, uint32_t cbw) {
        u64 key = 0;
        if (owner && fold) {                       // step p's duty cycle first: float32, two separately rounded operations
            duty = duty * d.mom;
            if ((cbw >> (row & 31)) & 1u) {
                if (fold == 2) owner = false;
                else duty = duty + d.dinc;
            }
            if (owner) d.duty[row] = duty;
        }
        if (owner) {
            d.overlap[sp][row] = cn;
            const float f = htm_exp_f32(d.coef * duty);                // float32 product, documented exp
            const double bo = (double)f * (double)cn;                  // exact (24-bit x <= 16-bit)
            d.boosted[sp][row] = bo;
            key = select_key(bo);
            d.key[sp][row] = key;
        }
        // (plain LDS atomics: hist_add's loop over the distinct digits, a dependent shuffle + ballot + atomic each, cost
        // 0.3 us per call here)
        // (the windowed histogram leaves out the bin below the window, where most columns are: the select counts from the
        // top and never gets there -- if it would, the k-th key is outside the window and the exact fallback takes over)
        const uint32_t bin = wmode ? win_bin(key, wbase) : (uint32_t)(key >> sel_shift(0));
        if (owner && (!wmode || bin != 0u)) {
            atomicAdd(&h[bin], 1u);
            if (wmode) atomicAdd(&hc[bin >> 6], 1u);
        }
    };
    for (int row0 = first_row0; row0 < d.c1; row0 += nwaves * rpw * U) {
        const bool first = row0 == first_row0;
        int cnt[U];
        float dty = dty_first;                     // (spread: fetched with the mask rows, not after the reduction)
        uint32_t cbw = cbw_first;
#pragma unroll
        for (int u = 0; u < U; ++u) cnt[u] = 0;
        if (!first && spread) {
            const int row = row0 + (l & 3) * rpw + sub;
            dty = (l < U && row < d.c1) ? d.duty[row] : 0.f;
            if (fold) cbw = (l < U && row < d.c1) ? d.colbits[p][row >> 5] : 0u;
        }
        for (int j = l; j < d.W4; j += G) {
            if (!(first && j == l)) {              // the mask rows first: they do not wait for the step counter
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int row = row0 + u * rpw + sub;
                    m[u] = row < d.c1 ? mask4[(uint32_t)row * (uint32_t)d.W4 + (uint32_t)j] : make_uint4(0, 0, 0, 0);
                }
            }
            const uint4 x = in4[j];
#pragma unroll
            for (int u = 0; u < U; ++u)
                cnt[u] += __popc(m[u].x & x.x) + __popc(m[u].y & x.y) + __popc(m[u].z & x.z) + __popc(m[u].w & x.w);
        }
        OV_STAMP(2);
        if (spread) {
            int mine = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cn = group8_sum_all(cnt[u]);
                if (l == u) mine = cn;
            }
            const int row = row0 + (l & 3) * rpw + sub;
            finish_row(l < U && row < d.c1, row, mine, dty, cbw);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int cn = cnt[u];
                for (int o = G >> 1; o > 0; o >>= 1) cn += __shfl_xor(cn, o);
                const int row = row0 + u * rpw + sub;
                const bool mine = l == 0 && row < d.c1;
                finish_row(mine, row, cn, mine ? d.duty[row] : 0.f, (fold && mine) ? d.colbits[p][row >> 5] : 0u);
            }
        }
    }
    OV_STAMP(3);
    lds_barrier();                                 // (LDS only: the flush does not wait for the stores of the keys)
    OV_STAMP(4);
    uint32_t *g0 = d.hist0 + (size_t)sp * HIST0_PAR + (size_t)(blk & (HIST_REP - 1)) * SEL_BINS;
    uint32_t *gc = d.hist0 + (size_t)sp * HIST0_PAR + HIST0_FINE + (size_t)(blk & (COARSE_REP - 1)) * COARSE_STRIDE;
    const int n_fine = wmode ? SEL_BINS - SEL_COARSE : SEL_BINS;
    {   // (the block's bins first, all of them at once, then the atomics of those that hold anything: one LDS round trip)
        constexpr int PER = SEL_BINS / BS;
        uint32_t v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) v[k] = h[threadIdx.x + k * BS];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * BS;
            if (v[k] && i < n_fine) atomicAdd(&g0[i], v[k]);
        }
    }
    // the runs' sums were counted beside the bins (in the words of the LDS histogram the window never uses: bins >= WIN_BINS)
    if (wmode && threadIdx.x < SEL_COARSE && hc[threadIdx.x]) atomicAdd(&gc[threadIdx.x], hc[threadIdx.x]);
    OV_STAMP(5);
}

__global__ __launch_bounds__(RB) void k_sp_overlap(Dev d, const uint32_t *__restrict__ bank, int n_inputs, int G, int p, int sp, int step_offset, int wmode) {
    __shared__ uint32_t h[SEL_BINS];               // histogram of the top key digit (select pass 0), or the windowed one
    role_overlap<RB>(d, bank, n_inputs, G, p, sp, step_offset, blockIdx.x, gridDim.x, h, wmode);
}

// The same for ONE input that comes from the host with the launch itself (htm_step and its kin, input_dim <= 2048): the packed
// input travels in the kernel's arguments -- no copy to stage, nothing to wait for -- every block reads it from there (through
// LDS), and block 0 leaves it in d.input_stage for the launches that follow (the permanence rows of the step).
#define ARG_INPUT_WORDS 64
struct PackedInputArg { uint32_t w[ARG_INPUT_WORDS]; };
__global__ __launch_bounds__(RB) void k_sp_overlap_arg(Dev d, PackedInputArg in, int G, int p, int wmode) {
    __shared__ uint32_t h[SEL_BINS];
    __shared__ __attribute__((aligned(16))) uint32_t s_in[ARG_INPUT_WORDS];
    if (threadIdx.x < ARG_INPUT_WORDS) {
        const uint32_t v = threadIdx.x < d.W ? in.w[threadIdx.x] : 0u;
        s_in[threadIdx.x] = v;
        if (blockIdx.x == 0 && threadIdx.x < d.W) d.input_stage[threadIdx.x] = v;
    }
    __syncthreads();
    role_overlap<RB>(d, s_in, 1, G, p, p, 0, blockIdx.x, gridDim.x, h, wmode);
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
            const uint32_t *g0 = d.hist0 + (size_t)sp * HIST0_PAR;
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
    const uint32_t pre = wave_incl_scan(cs);      // inclusive prefix inside the wave; suffixes follow from the total
    const uint32_t wtot = (uint32_t)__builtin_amdgcn_readlane((int)pre, 63);
    if (lane == 0) s_wave[wv] = wtot;             // wave total
    __syncthreads();
    uint32_t above = wtot - pre;                  // keys in bins above my chunk, inside my wave ...
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
    for (int c0 = d.sel_lo + blk * BS + (tid & ~63); c0 < d.sel_hi; c0 += nblk * BS) {
        const int c = c0 + lane_id();
        const u64 key = c < d.sel_hi ? keys[c] : 0;
        // digits below the top one are spread over the bins: plain LDS atomics (hist_add's loop runs once per
        // distinct digit of the wave, which here is most of its lanes)
        if (c < d.sel_hi && ((key ^ prefix) & himask) == 0) atomicAdd(&sh->h[(uint32_t)(key >> shift) & (nb - 1)], 1u);
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
        d.ctr->sel_win[sp ^ 1] = min(win_base_for(T) + (uint32_t)d.win_offset, 4096u - WIN_COARSE);
    }
    if (d.sel_passes > 1)                       // pass-0 histogram is consumed: clear it for its next use
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HIST0_PAR; i += gridDim.x * 256) d.hist0[(size_t)sp * HIST0_PAR + i] = 0;
    const int c = d.sel_lo + blockIdx.x * 256 + threadIdx.x;
    uint32_t v = 0;
    if (c < d.sel_hi) {
        u64 key = d.key[sp][c];
        v = (key > T) ? 1u : ((key == T) ? 0x10000u : 0u);
    }
    uint32_t total;
    block_excl_scan<256>(v, s_wave, total);
    if (threadIdx.x == 0) d.sel_blk[blockIdx.x] = total;
}

// TemporalMemory.process up to the winner cells (networks.py:95-104) for ONE active column, executed by the KP lanes of
// a lane group (lane j of the group = cell j): a half-wave where cell_dim <= 32 (two columns per wave), the whole wave for
// cell_dim up to 64.  Bursting, best-matching cell (networks.py:73-82), least-used cell (:84-89).
// idx = position of column a in the ascending active list.  The words are 64 bits wide throughout; with 32 cell slots the
// upper halves are zero.
struct ColumnWords { u64 act, winner, unacc; bool burst; };

// what a lane (cell j of column a) reads from memory for tm_column_words: fetched for several columns at once where a
// lane group handles more than one (the loads of all of them in flight together)
struct ColumnLoads { float cm; int segcount; };

// op over the lanes of a group (a half-wave, or the wave when a column has 64 cell slots), the result in every lane of it
template <typename Op>
__device__ __forceinline__ float group_reduce(const Dev &d, float v, float ident, Op op) {
    const float r = half_reduce(v, ident, op);
    if (d.WPC == 1) return r;
    const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 0));
    const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 32));
    return op(lo, hi);
}
// the group's share of a ballot: its 32 bits, or all 64
__device__ __forceinline__ u64 group_ballot(const Dev &d, bool pred) {
    const u64 b = __ballot(pred);
    return d.WPC == 1 ? (u64)(uint32_t)(b >> (lane_id() & 32)) : b;
}

// the previous step's prediction words of column a (prev_state.cell_prediction row)
__device__ __forceinline__ u64 tm_pred_words(const Dev &d, int p, bool col_ok, int a) {
    if (!col_ok) return 0ull;
    if (d.WPC == 1) return d.pred[p ^ 1][a];
    return *(const u64 *)&d.pred[p ^ 1][2 * a];
}

__device__ __forceinline__ ColumnLoads tm_column_loads(const Dev &d, int p, bool col_ok, int a, int has_distal) {
    const int j = lane_id() & (d.KP - 1);
    const bool valid = col_ok && j < d.K;
    ColumnLoads l;
    const int idx = valid ? a * d.KP + j : 0;        // (clamped and unconditional: a guarded load is waited for on its own)
    const float cm = __uint_as_float(d.cellmax[p ^ 1][idx]);
    l.cm = (valid && has_distal) ? cm : -1.0f;
    l.segcount = d.segcount[idx];
    return l;
}

// pw = prev_state.cell_prediction row of column a (0 when !col_ok)
__device__ __forceinline__ ColumnWords tm_column_compute(const Dev &d, int want_winner, bool col_ok, int a, u64 pw, const ColumnLoads &l,
                                                         int has_distal, uint32_t step) {
    const int j = lane_id() & (d.KP - 1);
    const bool valid = col_ok && j < d.K;
    const bool burst = pw == 0;
    const u64 act = burst ? cell_mask64(d.K) : pw;           // networks.py:115
    const float cm = l.cm;
    u64 winner = pw, unacc = 0;
    if (want_winner) {
        const float colmax = group_reduce(d, cm, -3.0e38f, [](float x, float y) { return fmaxf(x, y); });
        const bool col_matching = has_distal && colmax >= (float)d.match_thr;      // networks.py:80
        const bool best = valid && has_distal && fabsf(cm - colmax) < d.eps;       // :81
        float jit = 3.0e38f;
        if (valid) {
            uint32_t base = htm_stream_base(d.seed, HTM_STREAM_LEAST_USED, step);
            jit = htm_jitter((float)l.segcount, htm_draw24(base, (uint32_t)(a * d.K + j), 0u));   // :86-87
        }
        const float mn = group_reduce(d, jit, 3.0e38f, [](float x, float y) { return fminf(x, y); });
        const bool least = valid && fabsf(jit - mn) < d.eps;                       // :88
        const bool wbit = col_matching ? best : least;
        const u64 pick = group_ballot(d, wbit);
        if (burst) winner = pick;                                                  // :102
        const u64 bm = group_ballot(d, valid && has_distal && !(cm < d.eps));      // cell has a matching segment
        unacc = has_distal ? (winner & ~bm) : 0ull;                                // projections.py:271
    }
    return ColumnWords{act, want_winner ? winner : 0ull, unacc, burst};
}

__device__ __forceinline__ ColumnWords tm_column_words(const Dev &d, int p, int want_winner, bool col_ok, int a, u64 pw) {
    const int has_distal = d.ctr->has_distal;
    const ColumnLoads l = tm_column_loads(d, p, col_ok, a, has_distal);
    return tm_column_compute(d, want_winner, col_ok, a, pw, l, has_distal, d.ctr->step[p]);
}

// store the words of active column a, the idx-th of the ascending active list: lane 32 * h of the group writes word h
// WT: the stores go through to memory (agent scope) -- the two-launch schedule, whose middle role reads them in the SAME
// launch, from whichever XCD its blocks run on (k_act_mid_rows)
template <bool WT = false, typename T>
__device__ __forceinline__ void tm_put(T *ptr, T v) {
    if (WT) __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *ptr = v;
}
template <bool WT = false>
__device__ __forceinline__ void tm_store_column(const Dev &d, int p, bool col_ok, int a, int idx, const ColumnWords &w) {
    const int j = lane_id() & (d.KP - 1);
    if (col_ok && (j & 31) == 0) {
        const int h = j >> 5, wi = a * d.WPC + h, s = idx * d.WPC + h;
        const uint32_t act = (uint32_t)(w.act >> (32 * h)), win = (uint32_t)(w.winner >> (32 * h));
        tm_put<WT>(&d.act[p][wi], act);
        tm_put<WT>(&d.win[p][wi], win);
        if (h == 0) tm_put<WT>(&d.bursting[idx], (uint8_t)(w.burst ? 1 : 0));
        tm_put<WT>(&d.actw_id[s], wi);
        tm_put<WT>(&d.unacc_word[s], (uint32_t)(w.unacc >> (32 * h)));
        tm_put<WT>(&d.winw_idx[s], win);
        tm_put<WT>(&d.actcnt[s], (uint8_t)__popc(act));
        tm_put<WT>(&d.act_list[s], act);
    }
}

template <bool WT = false>
__device__ __forceinline__ void tm_activate_column(const Dev &d, int p, int want_winner, bool col_ok, int a, int idx, u64 pw) {
    tm_store_column<WT>(d, p, col_ok, a, idx, tm_column_words(d, p, want_winner, col_ok, a, pw));
}

// the active column a lane group of a 256-thread block takes: groups per block = 256 / KP, this thread's = tid / KP
__device__ __forceinline__ int tm_group_of(const Dev &d, int tid) { return tid >> d.LK; }
__device__ __forceinline__ int tm_groups_per_block(const Dev &d) { return 256 >> d.LK; }

// ---- column sharding: the exchange record ----------------------------------------------------
// One fixed-size record per rank and timestep (oracle/sharded.py record_nbytes): the rank's own
// candidates -- a superset of its top-min(k, own columns), n of them, n <= CAP slots --, in ascending column order, each
// with the cell words the column WOULD have if it became active (they only depend on the owner's previous predictions,
// segment maxima and segment counts), and the global ids of own segments that fell below the matching threshold while
// learning (the lowest-id-first recycling rule of projections.py:80-81 is global).  CAP = candidate slots per rank:
//   [boosted f64 x CAP][column | bursting << 31  u32 x CAP][winner word u32 x CAP][needs-a-segment word u32 x CAP]
//   [n_dead u32][dead ids u32 x DEAD_CAP][n u32][n_hot u32][hot floor: select key, u32 x 2], padded to 8 bytes
//   [hot boosted f64 x CAP][hot slot u16 x CAP], padded to 16 bytes
// (the active-cell word is not sent: it is all cells of a bursting column and the winner word otherwise; the boosted
// overlaps of the CAP - n unused slots are CAND_PAD, all bits set: the global select tells a candidate from a free slot by
// the value it loads anyway)
// The hot list: the rank's best few candidates -- every own key from the histogram bin of its hot_target-th largest on;
// the bin's lower edge is the list's floor --, their boosted overlaps once more, side by side, with their slots,
// ascending too.  A tenth of a step's candidates can win.  The global select takes the k-th largest of the ranks' hot keys
// (hot_target x ranks >= k); if that key is at or above every rank's floor, every candidate at or above it is in a hot list:
// it is the k-th of all, and nothing else of the records is read (a few KB instead of 100: one block's loads are bound
// by its CU).  n_hot = CAND_HOT_NONE: the rank has no hot list this step (its local select cut its threshold bin exactly, or
// the list would not fit).
#define CAND_PAD (~0ull)
#define CAND_HOT_NONE 0xFFFFFFFFu
__host__ __device__ __forceinline__ size_t shard_record_count_offset(int cap) { return (size_t)cap * 20 + 4 + 4 * DEAD_CAP; }
__host__ __device__ __forceinline__ size_t shard_record_hot_offset(int cap) { return (shard_record_count_offset(cap) + 16 + 7) / 8 * 8; }
__host__ __device__ __forceinline__ size_t shard_record_bytes(int cap) {
    size_t n = shard_record_hot_offset(cap) + (size_t)cap * 10;
    return (n + 15) / 16 * 16;
}
// candidate slots for a rank that must offer n_cand of n_local columns: room for the threshold bin of the windowed
// histogram on top.  The smaller the share of its columns a rank offers, the closer its cut lies to the ties of a learned
// pattern's columns and the fuller its bin: a quarter more slots where a rank offers more than an eighth of its columns
// (8 shards of 65 536 columns, 1 311 of 8 192: 13 steps of 1 000 cut exactly all the same), half as many again where it
// offers less (2 shards, 1 311 of 32 768: 93 steps with a quarter, 18 with a half)
__host__ __device__ __forceinline__ int shard_cand_cap(int n_cand, int n_local) {
    const int part = (long long)n_cand * 8 > n_local ? n_cand / 4 : n_cand / 2;
    const int slack = part > 64 ? part : 64;
    return n_cand + slack < n_local ? n_cand + slack : n_local;
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
#define CAND_D 15             // distinct bucket keys one block can publish (head + 15 granules = its 128-byte record)
#define CAND_RAW 64           // ... and collect from its waves before merging duplicates
#define CAND_MAX 1024         // bucket entries a block can merge
#define CAND_OTHERS 160         // ... after folding the copies of one key, if at most this many others remain (<= 256: one per thread)
#define CAND_PAIRWISE 160      // ... by comparing all pairs; above that, by radix refinement in LDS
#define ZOOM_BINS 1024         // more pairs than CAND_PAIRWISE: the sub-bin of the k-th key first (the next 10 key bits), then its pairs only

// pick the bucket that contains the krem-th largest key of a histogram held in LDS
// (bins [0, nb)); all BS threads call; returns bucket and the keys above it
template <int BS, bool LDS_ONLY = false>
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
    const uint32_t pre = wave_incl_scan(cs);
    const uint32_t wtot = (uint32_t)__builtin_amdgcn_readlane((int)pre, 63);
    if (lane == 0) s_wave[wv] = wtot;
    if (LDS_ONLY) lds_barrier(); else __syncthreads();
    uint32_t above = wtot - pre;
    for (int w = wv + 1; w < BS / 64; ++w) above += s_wave[w];
    if (above < krem && krem <= above + cs) {
        for (int b = min((tid + 1) * PER, nb) - 1; b >= tid * PER; --b) {
            const uint32_t hb = h[b];
            if (above + hb >= krem) { s_out[0] = (uint32_t)b; s_out[1] = above; break; }
            above += hb;
        }
    }
    if (LDS_ONLY) lds_barrier(); else __syncthreads();
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
#define EMIT_LOCAL 8            // a shard's own candidates: the list goes to cand_cols and, with each candidate's speculative
                                // cell words, into the exchange record (send); nothing else is touched
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
    uint32_t gt, eq, out[8], flags, krem, r;
    int n, nraw, ne;
};

// wmode (with fused): the launched part of the select was the windowed histogram (win_bin) of role_overlap
// diagnostic build (-DBITHTM_EMIT_STAMPS, handle created under BITHTM_TRACE=1): device clock at the phases of every emit
// block of the three-launch schedule, d.trace[block * 8 + phase], + the merged bucket entries (tools/emit_phases.py)
#ifdef BITHTM_EMIT_STAMPS
#define EMIT_STAMP(i) do { if (d.trace && wmode && tid == 0 && b < 1024) d.trace[(size_t)b * 8 + (i)] = wall_clock64(); } while (0)
#else
#define EMIT_STAMP(i) do { } while (0)
#endif
__device__ __forceinline__ void role_emit(const Dev &d, int p, int want_winner, int fused, int mode, int b, int nblk, EmitShared *sh, int wmode = 0) {
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
    uint32_t *s_zh = h + 2 * CAND_MAX;              // [ZOOM_BINS] the pairs' multiplicities by sub-bin, behind the merged keys
    static_assert(2 * CAND_MAX + ZOOM_BINS <= SEL_BINS, "bucket keys and their sub-bins must fit the histogram");
    const int tid = threadIdx.x, lane = lane_id();
    EMIT_STAMP(0);
    if (tid == 0) { s_gt = 0; s_eq = 0; s_n = 0; s_nraw = 0; s_ne = 0; s_flags = 0; s_T = 0; }
    const int cbase = d.sel_lo + b * 256, c = cbase + tid;      // the select covers columns [sel_lo, sel_hi)
    const bool local = mode & EMIT_LOCAL;
    // independent of everything below: in flight while the select state is resolved
    const u64 my_key = c < d.sel_hi ? d.key[p][c] : 0;
    const bool own_col = c < d.sel_hi && c >= d.c0 && c < d.c1;
    const float my_duty = (own_col && (mode & EMIT_DUTY)) ? d.duty[c] : 0.f;
    const bool tm_here = d.act[0] && (mode & EMIT_ACTIVATE);
    s_predw[tid] = (c < d.sel_hi && (tm_here || local) && d.WPC == 1) ? d.pred[p ^ 1][c] : 0u;
    u64 T;
    uint32_t r;                                     // how many of the keys == T are selected
    bool second_round = false;                      // per-block counts still to be exchanged
    const uint32_t epoch = (d.ctr->emit_epoch & 0x3FFu) + 1u;           // 1..1024, changes with every overlap launch
    if (fused && wmode && tid >= 64)                // (the sub-bin counts of a crowded bin, below: zeroed by the waves that wait while
        for (int i = tid - 64; i < ZOOM_BINS; i += 192) s_zh[i] = 0;      // the first one resolves the window's sums)
    if (fused) {
        u64 P;
        uint32_t krem;
        int lowbits = sel_shift(d.sel_passes - 1);              // key bits not resolved by launches
        bool take_all = false;                                  // (a shard's candidates: the whole threshold bin fits the record)
        uint32_t n_all = 0;
        bool hot_ok = false;                                    // (... and the rank's best few fit its hot list)
        u64 P_hot = 0;                                          // the lowest key a hot candidate can have
        if (wmode) {
            // The bin of the k-th key, and the keys above it: from the runs' sums first (64 of them, one wave, one read of the
            // copies), then from the 64 bins of the chosen run -- two dependent reads of a kilobyte each, no block-wide scan.
            // (Round 2 had every block fetch the copies whole, 64 KB, and scan 4 096 bins: 3.0-4.5 us at the head of the
            // select finish's chain.)
            // (a shard's candidates: the second wave does the same for the hot_target-th key, the floor of the rank's hot list)
            if (tid < 64 || (local && tid < 128)) {
                const uint32_t *g0 = d.hist0 + (size_t)p * HIST0_PAR;
                const int idx = 63 - (tid & 63);     // (from the top: the scan then gives "keys above")
                uint32_t v = 0;
#pragma unroll
                for (int r = 0; r < COARSE_REP; ++r) v += g0[HIST0_FINE + r * COARSE_STRIDE + idx];
                uint32_t incl = wave_incl_scan(v);
                const uint32_t kk = tid < 64 ? (uint32_t)d.sel_k : (uint32_t)d.hot_target;
                uint32_t *s_out = sh->out + (tid < 64 ? 0 : 4);
                const u64 hit = __ballot(incl >= kk);          // the first run (from the top) at which k keys have been seen
                const int lane_r = hit ? __ffsll((long long)hit) - 1 : 63;
                const int run = 63 - lane_r;
                const uint32_t above_run = (uint32_t)__builtin_amdgcn_readlane((int)(incl - v), lane_r);
                uint32_t f = 0;
#pragma unroll
                for (int r = 0; r < HIST_REP; ++r) f += g0[(size_t)r * SEL_BINS + run * 64 + idx];
                const uint32_t incl_f = above_run + wave_incl_scan(f);
                const u64 hit_f = __ballot(incl_f >= kk);
                const int lane_f = hit_f ? __ffsll((long long)hit_f) - 1 : 63;
                if ((tid & 63) == lane_f) { s_out[0] = hit && hit_f ? (uint32_t)(run * 64 + 63 - lane_f) : 0u; s_out[1] = incl_f - f; s_out[2] = f; }
            }
            __syncthreads();
            const uint32_t bucket = s_out[0], above = s_out[1], in_bin = s_out[2];
            const uint32_t bucket_hot = s_out[4], n_hot_all = s_out[5] + s_out[6];
            __syncthreads();
            take_all = local && d.cand_take_all && bucket >= 1u && bucket <= (WIN_COARSE << WIN_FINE) && above + in_bin <= (uint32_t)d.cand_cap;
            n_all = above + in_bin;
            // the hot list: the keys from the bin of the hot_target-th on (a bin of the window or the one above it), if they fit
            hot_ok = take_all && bucket_hot >= bucket && n_hot_all <= (uint32_t)d.hot_budget && d.cand_cap <= 65536;
            if (hot_ok) {
                const uint32_t fine = bucket_hot - 1u;
                P_hot = ((u64)(d.ctr->sel_win[p] + (fine >> WIN_FINE)) << 52) | ((u64)(fine & ((1u << WIN_FINE) - 1u)) << WIN_LOWBITS);
            }
            if (bucket >= 1u && bucket <= (WIN_COARSE << WIN_FINE)) {
                const uint32_t fine = bucket - 1u;
                P = ((u64)(d.ctr->sel_win[p] + (fine >> WIN_FINE)) << 52) | ((u64)(fine & ((1u << WIN_FINE) - 1u)) << WIN_LOWBITS);
                krem = (uint32_t)d.sel_k - above;
                lowbits = WIN_LOWBITS;
            } else {                                 // the k-th key is outside the window: nothing resolved, the fallback does it all
                P = 0;
                krem = (uint32_t)d.sel_k;
                lowbits = 64;
            }
        } else {
            sel_resolve<256>(d, p, d.sel_passes - 1, h, s_wave, &P, &krem, &s_prefix, &s_krem);    // (ends behind a barrier)
        }
        EMIT_STAMP(1);                               // (the launched part of the select is resolved)
        const u64 hiP = lowbits < 64 ? P >> lowbits : 0ull, hi = lowbits < 64 ? my_key >> lowbits : 0ull;
        if (take_all) {
            // A shard's candidates only have to CONTAIN its top-min(k, own columns): when the bins above the threshold bin and the
            // bin itself fit the record's slots -- every block sees that in the histogram, so all decide alike -- every key of
            // the bin goes in and nothing is left to resolve: no record exchange, no k-th key.  What the blocks still owe each
            // other is a count (where in the record a block's candidates start, and where its hot ones do), published first and
            // read last: the cell words of the block's candidates are computed in between.
            const bool sel = c < d.sel_hi && hi >= hiP;
            const bool hot = sel && hot_ok && my_key >= P_hot;
            uint32_t total;
            const uint32_t ex = block_excl_scan<256, true>((sel ? 1u : 0u) | (hot ? 0x10000u : 0u), s_wave, total);
            const int n_sel = (int)(total & 0xFFFFu), n_hot_blk = (int)(total >> 16);
            const uint32_t tag2 = epoch | 0x800u;
            if (tid == 0)
                __hip_atomic_store(&d.sel_blk[b], (tag2 << 20) | ((uint32_t)n_hot_blk << 10) | (uint32_t)n_sel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t *s_cw = h, *s_wn = h + 256, *s_un = h + 512;      // (the histogram's LDS is free on this path)
            double *s_bo = (double *)(h + 768);
            int *s_hot = (int *)(h + 1280);
            if (sel) s_col[ex & 0xFFFFu] = c;
            if (hot) s_hot[ex >> 16] = (int)(ex & 0xFFFFu);
            lds_barrier();                       // (LDS only: not the published count's round trip, not the loads in flight)
            EMIT_STAMP(2);
            {
                const int has_distal = d.ctr->has_distal;
                const uint32_t step = d.ctr->step[p];
                constexpr int CPP = 8;           // (64 candidates per pass: a block has 40-50, all their loads in flight at once)
                for (int i0 = 0; i0 < n_sel; i0 += 8 * CPP) {
                    int a[CPP];
                    bool ok[CPP];
                    ColumnLoads l[CPP];
                    double bo[CPP];
#pragma unroll
                    for (int u = 0; u < CPP; ++u) {
                        const int i = i0 + u * 8 + (tid >> 5);
                        ok[u] = i < n_sel;
                        a[u] = ok[u] ? s_col[i] : d.c0;
                        l[u] = tm_column_loads(d, p, ok[u], a[u], has_distal);
                        bo[u] = d.boosted[p][a[u]];
                    }
#pragma unroll
                    for (int u = 0; u < CPP; ++u) {
                        const ColumnWords w = tm_column_compute(d, 1, ok[u], a[u], ok[u] ? s_predw[a[u] - cbase] : 0u, l[u], has_distal, step);
                        const int i = i0 + u * 8 + (tid >> 5);
                        if (ok[u] && (lane & 31) == 0) {
                            s_bo[i] = bo[u];
                            s_cw[i] = (uint32_t)a[u] | (w.burst ? 0x80000000u : 0u);
                            s_wn[i] = w.winner;
                            s_un[i] = w.unacc;
                        }
                    }
                }
            }
            // everybody's counts (all of them, not just the earlier blocks': a block that has published has read the histogram,
            // which is cleared below; and the hot lists' total decides where the padding starts)
            uint32_t before = 0, before_hot = 0, all_hot = 0;
            for (int i = tid; i < nblk; i += 256) {
                uint32_t v = 0;
                int spins = 0;
                do {
                    v = __hip_atomic_load(&d.sel_blk[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((v >> 20) == tag2) break;
                    __builtin_amdgcn_s_sleep(1);
                } while (++spins < (1 << 22));
                if ((v >> 20) != tag2) atomicOr(&d.ctr->error, 16);
                if (i < b) { before += v & 0x3FFu; before_hot += (v >> 10) & 0x3FFu; }
                all_hot += (v >> 10) & 0x3FFu;
            }
            before = wave_sum(before);
            before_hot = wave_sum(before_hot);
            all_hot = wave_sum(all_hot);
            if (lane == 0) {
                if (before) atomicAdd(&s_gt, before);
                if (before_hot) atomicAdd(&s_eq, before_hot);
                if (all_hot) atomicAdd(&s_n, (int)all_hot);
            }
            lds_barrier();
            EMIT_STAMP(3);
            const int first_pos = (int)s_gt, first_hot = (int)s_eq, n_hot = s_n;
            {
                const int cap = d.cand_cap;
                double *r_boost = (double *)d.send;
                uint32_t *r_col = (uint32_t *)(d.send + (size_t)cap * 8), *r_win = r_col + cap, *r_unacc = r_win + cap;
                double *r_hot = (double *)(d.send + shard_record_hot_offset(cap));
                uint16_t *r_hot_slot = (uint16_t *)(r_hot + cap);
                if (tid < n_sel && first_pos + tid < cap) {
                    const int pos = first_pos + tid;
                    r_boost[pos] = s_bo[tid];
                    r_col[pos] = s_cw[tid];
                    r_win[pos] = s_wn[tid];
                    r_unacc[pos] = s_un[tid];
                }
                if (tid < n_hot_blk && first_hot + tid < cap) {
                    const int li = s_hot[tid];
                    r_hot[first_hot + tid] = s_bo[li];
                    r_hot_slot[first_hot + tid] = (uint16_t)(first_pos + li);
                }
                for (int i = (int)n_all + b * 256 + tid; i < cap; i += nblk * 256) ((u64 *)r_boost)[i] = CAND_PAD;
                for (int i = n_hot + b * 256 + tid; i < cap; i += nblk * 256) ((u64 *)r_hot)[i] = CAND_PAD;
                if (b == 0) {
                    uint32_t *r_dead = r_unacc + cap;
                    const int n = min(d.dead_list[0], DEAD_CAP);
                    if (tid == 0) {
                        r_dead[0] = (uint32_t)n;
                        uint32_t *r_n = (uint32_t *)(d.send + shard_record_count_offset(cap));
                        r_n[0] = n_all;
                        r_n[1] = hot_ok ? (uint32_t)n_hot : CAND_HOT_NONE;
                        r_n[2] = (uint32_t)P_hot;                                      // (the list's floor: every own key at or above it is in the list)
                        r_n[3] = (uint32_t)(P_hot >> 32);
                        // (the coming step's window around this bin; the k-th key itself is not known on this path)
                        d.ctr->sel_win[p ^ 1] = min(win_base_for(P) + (uint32_t)d.win_offset, 4096u - WIN_COARSE);
                    }
                    for (int j = tid; j < n; j += 256) r_dead[1 + j] = (uint32_t)d.dead_list[1 + j];
                }
            }
            if (d.sel_passes > 1)
                for (int i = b * 256 + tid; i < HIST0_PAR; i += nblk * 256) d.hist0[(size_t)p * HIST0_PAR + i] = 0;
            EMIT_STAMP(4);
            EMIT_STAMP(5);
            return;
        }
        const bool c_gt = c < d.sel_hi && hi > hiP, c_cand = c < d.sel_hi && hi == hiP;
        if (!wmode)
            for (int i = tid; i < ZOOM_BINS; i += 256) s_zh[i] = 0;  // (the resolved histogram's LDS is free from here on)
        // ---- this block's record
        {
            const u64 mg = __ballot(c_gt);
            if (lane == 0 && mg) atomicAdd(&s_gt, (uint32_t)__popcll(mg));        // s_gt: keys above the bucket, for now
            u64 todo = __ballot(c_cand);
            while (todo) {                           // group equal bucket keys inside the wave
                const int leader = __ffsll((long long)todo) - 1;
                const u64 kl = wave_read(my_key, leader);
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
        int first = -1;                              // merge duplicates that came from different waves: the first wave alone, its
        if (tid < 64 && nraw > 1) {                  // lanes holding a raw key each (no walk along dependent LDS reads, no block barrier)
            const u64 k = lane < nraw ? s_bk[lane] : 0ull;
            const uint32_t cn = lane < nraw ? s_bc[lane] : 0u;
            for (int j = 0; j < nraw; ++j) {         // (uniform)
                const u64 kj = wave_read(k, j);
                if (first < 0 && lane < nraw && kj == k) first = j;
            }
            if (lane < nraw && first != lane) atomicAdd(&s_bc[first], cn);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (read again below, by this wave)
        } else if (tid < nraw) {
            first = tid;
        }
        // record = up to 16 self-validating 64-bit granules (form R2: every granule carries the epoch, one
        // aligned 8-byte write-through store each, so no separate tag and no drain):
        //   [0]      epoch:12 | overflow:1 | pairs:4 | keys above the bucket:9 | multiplicity:9 | key bits:29
        //   [j >= 1] epoch:12 | multiplicity:12 | the unresolved key bits above low_zero, at most 40   (the high bits are the bucket's)
        // The first pair rides in the head granule -- the low_zero bottom bits of every key are zero, so
        // 29 bits hold the rest for input_dim up to 2^17 -- and a block with at most one bucket key, the
        // usual case and the one of a many-way tie, is read with a single load.
        u64 *rec = (u64 *)(d.sel_rec + (size_t)b * 32);
        const u64 etag = (u64)epoch << 52;
        const u64 lowmask = lowbits < 64 ? (1ull << lowbits) - 1ull : ~0ull;
        const bool inline_ok = lowbits - d.low_zero <= 29;
        if (tid < 64) {                              // wave 0 compacts the survivors into the record
            const bool alive = tid < nraw && first == tid;
            const u64 ma = __ballot(alive);
            const int n_pairs = __popcll(ma), pos = __popcll(ma & lanemask_lt());
            const bool overflow = s_nraw > CAND_RAW || n_pairs > d.cand_d || (n_pairs > 0 && !inline_ok) || lowbits - d.low_zero > 40;
            const u64 mine = alive ? (s_bk[tid] & lowmask) : 0ull;
            const uint32_t cnt = alive ? s_bc[tid] : 0u;
            if (alive && pos >= 1 && pos < CAND_D)
                __hip_atomic_store(rec + pos, etag | ((u64)cnt << 40) | (mine >> d.low_zero), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int l0 = ma ? __ffsll((long long)ma) - 1 : 0;      // the lane of pair 0
            const u64 k0 = wave_read(mine, l0);
            const uint32_t c0 = wave_read(cnt, l0);
            if (tid == 0) {
                const u64 pair0 = (ma && inline_ok) ? (((u64)(c0 & 0x1FFu) << 29) | (k0 >> d.low_zero)) : 0ull;
                __hip_atomic_store(rec, etag | (overflow ? (1ull << 51) : 0ull) | ((u64)min(n_pairs, CAND_D) << 47) | ((u64)my_gt_hi << 38) | pair0,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (tid == 0) s_gt = 0;
        EMIT_STAMP(2);                               // (own record published)
        // ---- everybody's records: the head granule is polled alone (one lane-load per spin keeps the
        // polling traffic low); further pairs, if any, are fetched in one batch; each granule validates itself
        // Not yet: a look that comes before the last blocks have published costs a whole round trip -- and 65 536 such loads (256
        // blocks x 256 records) in the way of the roles that share the launch: with the first look 0.8 us later the learning role
        // and the scan run beside less of that, and the third launch's median goes from 8.9 to 8.0 us (BITHTM_POLL_DELAY, in
        // units of 4 x 64 clocks: 4 = 0.45 us ... 8 = 0.9 us all within a percent; 0 and 16 both 4 % slower).
        if (nblk >= 32)
            for (int i = 0; i < d.poll_delay; ++i) __builtin_amdgcn_s_sleep(4);
        uint32_t gthi_before = 0;
        u64 heaviest = 0;                            // multiplicity:12 | unresolved key bits:40 of the heaviest pair this thread has read
        // (... and every pair's multiplicity into the histogram of the next ZB key bits: for a crowded bin, below)
        const int sigbits = lowbits - d.low_zero;                      // unresolved key bits that can be set (<= 40 unless a record says overflow)
        const int ZB = sigbits >= 10 ? 10 : (sigbits > 0 ? sigbits : 0), zsh = sigbits > ZB ? sigbits - ZB : 0;
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
                for (int j = 1; j < CAND_D; ++j) g[j] = j < np ? __hip_atomic_load(rr + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                bool ok = true;
#pragma unroll
                for (int j = 1; j < CAND_D; ++j) ok = ok && (j >= np || (g[j] >> 52) == epoch);
                if (ok) break;
                if (spins >= (1 << 20)) { atomicOr(&d.ctr->error, 16); np = 0; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (rb < b) gthi_before += (uint32_t)((g[0] >> 38) & 0x1FFu);
            if ((g[0] >> 51) & 1ull) atomicOr(&s_flags, 1u);
            // the pairs into the merged list: ONE reservation per wave for all its records' pairs (a reservation per lane is a
            // same-address LDS atomic served lane by lane, one per wave and pair index a chain of up to 9 returning atomics in a
            // crowded bin: 400-570 pairs, microseconds either way)
            // (the lanes here are the wave's first n -- tid < nblk in the last round --: the scan's sources are lower lanes, and
            // the total is the last of them's)
            const uint32_t incl = wave_incl_scan((uint32_t)np);
            const uint32_t wtot = wave_read(incl, 63 - __clzll((long long)__ballot(true)));
            if (wtot) {                               // (wave-uniform)
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_ne, (int)wtot);
                const int slot0 = wave_read(base, 0) + (int)(incl - (uint32_t)np);
#pragma unroll
                for (int j = 0; j < CAND_D; ++j) {
                    if (!__ballot(j < np)) break;
                    if (j < np && slot0 + j < CAND_MAX) {
                        const u64 low40 = j == 0 ? g[0] & 0x1FFFFFFFull : g[j] & 0xFFFFFFFFFFull;
                        const u64 cnt = j == 0 ? (g[0] >> 29) & 0x1FFu : (g[j] >> 40) & 0xFFFu;
                        s_ek[slot0 + j] = (lowbits < 64 ? hiP << lowbits : 0ull) | (low40 << d.low_zero);
                        s_ec[slot0 + j] = (uint16_t)cnt;
                        s_eb[slot0 + j] = (uint16_t)rb;
                        heaviest = max(heaviest, (cnt << 40) | low40);
                        atomicAdd(&s_zh[(uint32_t)(low40 >> zsh) & (ZOOM_BINS - 1)], (uint32_t)cnt);
                    }
                }
            }
        }
        // (the heaviest pair of all: for the many-way tie below -- settled here, before the barrier everybody needs anyway)
        if (__any(heaviest != 0)) {
            heaviest = wave_reduce64(heaviest, 0ull, [](u64 a, u64 b) { return a > b ? a : b; });
            if (lane == 0) atomicMax((unsigned long long *)&s_T, heaviest);
        }
        __syncthreads();
        int ne = s_ne;
        EMIT_STAMP(3);                               // (everybody's records read)
#ifdef BITHTM_EMIT_STAMPS
        if (d.trace && wmode && tid == 0 && b < 1024) d.trace[(size_t)b * 8 + 7] = (unsigned long long)ne | ((unsigned long long)s_nraw << 32);
        if (d.trace && wmode && b == 0 && ne > 160)      // the merged entries themselves, behind the emit blocks' rows
            for (int e = tid; e < min(ne, 1024); e += 256) d.trace[(size_t)256 * 8 + e] = ((s_ek[e] & lowmask) >> d.low_zero) | ((unsigned long long)s_ec[e] << 32) | ((unsigned long long)s_eb[e] << 48);
#endif
        bool settled = false;                        // (the k-th key is known already: the sub-bin held one key)
        // (Only where blocks hold SEVERAL keys each -- more pairs than blocks by a quarter --: a many-way tie alone, one pair per
        // block, is settled by the fold's first pass a microsecond sooner than by the two stages here; measured both ways.)
        if (!(s_flags & 1u) && ne <= CAND_MAX && ne > d.cand_pairwise && ne > (d.cand_zoom >= 0 ? d.cand_zoom : nblk + (nblk >> 2)) && ZB > 0) {
            // A crowded bin -- a learned pattern's thousand columns behind ONE key and a few dozen stragglers a part in 10^5 off
            // it, 400-570 (key, block) pairs -- is not ranked whole: the sub-bin of the k-th key is picked from the histogram of
            // the next ZB key bits, and only THAT sub-bin's pairs stay on the list (one key, nearly always: then it IS the k-th
            // key and nothing is ranked; otherwise the stages below run on the short list); the pairs above it are winners outright.
            // (two barrier stages, not five: the entries go into registers first, the first wave picks alone)
            constexpr int EPT = CAND_MAX / 256;
            u64 ek[EPT];
            uint32_t ecb[EPT];
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int e = tid + 256 * u;
                ek[u] = e < ne ? s_ek[e] : 0ull;
                ecb[u] = e < ne ? (uint32_t)s_ec[e] | ((uint32_t)s_eb[e] << 16) : 0u;
            }
            if (tid < 64) {                          // lane l: sub-bins (nb - 16 (l + 1), nb - 16 l] from the top, then lane j: the j-th of
                constexpr int PER = ZOOM_BINS / 64;  // the chosen sixteen -- two reads, two scans, no walk along dependent reads
                const int top = (1 << ZB) - 1 - PER * lane;
                uint32_t cs = 0;
#pragma unroll
                for (int j = 0; j < PER; ++j) cs += top - j >= 0 ? s_zh[top - j] : 0u;
                const uint32_t incl = wave_incl_scan(cs);
                const u64 hit = __ballot(incl >= krem);
                const int hl = hit ? __ffsll((long long)hit) - 1 : 0;
                const uint32_t above_g = wave_read(incl - cs, hl);
                const int bin = (1 << ZB) - 1 - PER * hl - lane;
                const uint32_t f = lane < PER && bin >= 0 ? s_zh[bin] : 0u;
                const uint32_t incl_f = above_g + wave_incl_scan(f);
                const u64 hit_f = __ballot(lane < PER && incl_f >= krem);
                if (hit_f && lane == __ffsll((long long)hit_f) - 1) { s_out[0] = (uint32_t)bin; s_out[1] = incl_f - f; }
                // (the kept pairs are counted in s_nraw: a wave that has not read s_ne yet may still be behind the barrier above)
                if (tid == 0) { s_nraw = 0; s_prefix = 0; if (b == 0) d.ctr->sel_zooms += 1; }
            }
            lds_barrier();
            const uint32_t zF = s_out[0];
            krem -= s_out[1];
            u64 and_or = 0;                          // ~(and of the kept keys' low words) : or of them -- one key kept <=> the halves are complements
            u64 mk[EPT];
            int n_keep = 0;
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const bool valid = tid + 256 * u < ne;
                const u64 low40 = (ek[u] & lowmask) >> d.low_zero;
                const uint32_t sub = (uint32_t)(low40 >> zsh) & (ZOOM_BINS - 1);
                if (valid && sub > zF && (int)(ecb[u] >> 16) < b) gthi_before += ecb[u] & 0xFFFFu;
                mk[u] = __ballot(valid && sub == zF);
                n_keep += __popcll(mk[u]);
            }
            if (n_keep) {                            // (wave-uniform; one reservation for all the wave keeps)
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_nraw, n_keep);
                int slot = wave_read(base, 0);
#pragma unroll
                for (int u = 0; u < EPT; ++u) {
                    if ((mk[u] >> lane) & 1ull) {
                        const int at = slot + __popcll(mk[u] & lanemask_lt());
                        const u64 low40 = (ek[u] & lowmask) >> d.low_zero;
                        s_ek[at] = ek[u];
                        s_ec[at] = (uint16_t)(ecb[u] & 0xFFFFu);
                        s_eb[at] = (uint16_t)(ecb[u] >> 16);
                        if (at == 0) s_T = ((u64)(ecb[u] & 0xFFFFu) << 40) | low40;      // (the fold's first guess, should more than one key be left)
                        and_or |= ((u64)~(uint32_t)low40 << 32) | (u64)(uint32_t)low40;
                    }
                    slot += __popcll(mk[u]);
                }
                and_or = wave_reduce64(and_or, 0ull, [](u64 a, u64 b) { return a | b; });
                if (lane == 0) atomicOr((unsigned long long *)&s_prefix, and_or);
            }
            lds_barrier();
            ne = s_nraw;
            // (the kept keys agree above their low 32 bits -- the bin's prefix and the sub-bin, zsh <= 30 bits below it)
            if (ne > 0 && (uint32_t)s_prefix == ~(uint32_t)(s_prefix >> 32)) { T = s_ek[0]; r = krem; settled = true; }
        }
        if (!(s_flags & 1u) && ne <= CAND_MAX) {
            bool folded = settled;
            if (settled) {                              // (nothing left to rank)
            } else if (ne <= d.cand_pairwise) {         // the krem-th largest of the merged bucket: all pairs
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
                // many entries: overlaps tie and most blocks report the same key -- the columns of one learned pattern, equal
                // overlaps and equal duty histories, a thousand of them behind one key, and a few dozen stragglers around it.
                // K0 = the key of the heaviest pair any block published (the largest key among equally heavy ones: found while the
                // records were read).  One pass folds its copies, counts the keys above it and sets the others aside;
                // nearly always the k-th key IS K0 (it straddles the k-th position) and that pass is all there is to do.
                const u64 K0 = (lowbits < 64 ? hiP << lowbits : 0ull) | ((s_T & 0xFFFFFFFFFFull) << d.low_zero);
                if (tid == 0) { sh->n_others = 0; sh->c0 = 0; s_mh[1] = 0; s_mh[2] = 0; }
                __syncthreads();
                // (the keys above K0 are set aside from the front of the list, the keys below it from its end)
                uint32_t c0 = 0, g0 = 0;
                for (int e0 = tid & ~63; e0 < ne; e0 += 256) {        // (whole waves: the appends are reserved once per wave)
                    const int e = e0 + lane;
                    const u64 ke = e < ne ? s_ek[e] : K0;
                    const uint32_t ce = e < ne ? (uint32_t)s_ec[e] : 0u;
                    const bool up = ke > K0, dn = ke < K0;
                    c0 += ke == K0 ? ce : 0u;
                    g0 += up ? ce : 0u;
                    const u64 mu = __ballot(up), md = __ballot(dn);
                    int bu = 0, bd = 0;
                    if (lane == 0) {
                        if (mu) bu = atomicAdd(&sh->n_others, __popcll(mu));
                        if (md) bd = (int)atomicAdd(&s_mh[2], (uint32_t)__popcll(md));
                    }
                    bu = wave_read(bu, 0);
                    bd = wave_read(bd, 0);
                    if (up) {
                        const int pos = bu + __popcll(mu & lanemask_lt());
                        if (pos < CAND_OTHERS) { sh->ok[pos] = ke; sh->oc[pos] = ce; }
                    } else if (dn) {
                        const int pos = CAND_OTHERS - 1 - (bd + __popcll(md & lanemask_lt()));
                        if (pos >= 0) { sh->ok[pos] = ke; sh->oc[pos] = ce; }
                    }
                }
                c0 = wave_sum(c0);
                g0 = wave_sum(g0);
                if (lane == 0) { if (c0) atomicAdd(&sh->c0, c0); if (g0) atomicAdd(&s_mh[1], g0); }
                __syncthreads();
                const int n_above = sh->n_others, n_below = (int)s_mh[2];
                const uint32_t G = s_mh[1], E0 = sh->c0;
                if (d.cand_speculate && G < krem && krem <= G + E0) {         // the k-th key is K0
                    T = K0;
                    r = krem - G;
                    folded = true;
                } else if (d.cand_speculate && n_above + n_below <= d.cand_others) {
                    // the k-th key is one of the stragglers on ONE side of K0 (the pattern's columns fill most of the k places, the
                    // rest go to the best of the others): all pairs among that side's entries only
                    const bool up = krem <= G;
                    const int first = up ? 0 : CAND_OTHERS - n_below, n_side = up ? n_above : n_below;
                    const uint32_t kside = up ? krem : krem - G - E0;
                    for (int e = tid; e < n_side; e += 256) {
                        const u64 ke = sh->ok[first + e];
                        uint32_t ng = 0, nq = 0;
                        for (int f = 0; f < n_side; ++f) {
                            const u64 kf = sh->ok[first + f];
                            const uint32_t cf = sh->oc[first + f];
                            ng += kf > ke ? cf : 0u;
                            nq += kf == ke ? cf : 0u;
                        }
                        if (ng < kside && kside <= ng + nq) { s_T = ke; s_r = kside - ng; }
                    }
                    __syncthreads();
                    T = s_T;
                    r = s_r;
                    folded = true;
                } else {
                    // (BITHTM_CAND_SPECULATE=0, the form of round 2: K0 joins the others, all pairs among them all)
                    const int no = n_above + n_below;
                    folded = no < d.cand_others;
                    if (folded) {
                        // close the gap between the two ends (at most CAND_OTHERS <= 256 entries: one per thread, read before any is written)
                        const bool mv = tid < n_below;
                        const u64 kk = mv ? sh->ok[CAND_OTHERS - 1 - tid] : 0ull;
                        const uint32_t cc = mv ? sh->oc[CAND_OTHERS - 1 - tid] : 0u;
                        __syncthreads();
                        if (mv) { sh->ok[n_above + tid] = kk; sh->oc[n_above + tid] = cc; }
                        if (tid == 0) { sh->ok[no] = K0; sh->oc[no] = E0; }
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
                T = (lowbits < 64 ? hiP << lowbits : 0ull) | pref;
                r = rem;
            }
            EMIT_STAMP(6);                            // (the k-th key itself; what follows counts the earlier blocks' winners)
            uint32_t g = gthi_before, e2 = 0;         // winners of the blocks before this one
            for (int e = tid; e < ne; e += 256)
                if (s_eb[e] < b) {
                    if (s_ek[e] > T) g += s_ec[e];
                    else if (s_ek[e] == T) e2 += s_ec[e];
                }
            g = wave_sum(g);
            e2 = wave_sum(e2);
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
                for (int c0 = d.sel_lo + (tid & ~63); c0 < d.sel_hi; c0 += 256) {
                    const int cc = c0 + lane;
                    const u64 kk = cc < d.sel_hi ? keys[cc] : 0;
                    hist_add(h, (uint32_t)(kk >> shift) & (nb - 1), cc < d.sel_hi && (top >= 64 || ((kk ^ P2) >> top) == 0));
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
        if (b == 0 && tid == 0) { d.ctr->sel_prefix[p] = T; d.ctr->sel_krem[p] = r; d.ctr->sel_win[p ^ 1] = min(win_base_for(T) + (uint32_t)d.win_offset, 4096u - WIN_COARSE); }
        // the pass-0 histogram is consumed: clear it for its next use (here, not earlier: a barrier
        // waits for outstanding stores, and the record exchange above is the critical chain)
        if (d.sel_passes > 1)
            for (int i = b * 256 + tid; i < HIST0_PAR; i += nblk * 256) d.hist0[(size_t)p * HIST0_PAR + i] = 0;
    } else {
        T = d.ctr->sel_prefix[p];
        r = d.ctr->sel_krem[p];
    }
    lds_barrier();                                   // (LDS only: the histogram's clearing stores are not waited for)
    EMIT_STAMP(4);                                   // (the k-th key is known)
    uint32_t flag = 0;
    if (c < d.sel_hi) flag = (my_key > T) ? 1u : ((my_key == T) ? 0x10000u : 0u);
    uint32_t total;
    const uint32_t ex = block_excl_scan<256, true>(flag, s_wave, total);
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
        g = wave_sum(g);
        e = wave_sum(e);
        if (lane == 0) { atomicAdd(&s_gt, g); atomicAdd(&s_eq, e); }
    }
    lds_barrier();
    const uint32_t gt_before = s_gt, eq_before = s_eq;
    const uint32_t g_run = gt_before + (ex & 0xFFFFu), e_run = eq_before + (ex >> 16);
    const int first_pos = (int)(gt_before + min(eq_before, r));
    const bool sel_any = c < d.sel_hi && ((flag & 1u) || ((flag >> 16) && e_run < r));
    if (d.act[0] && !local) {
        const u64 mcol = __ballot(sel_any);
        if (lane_id() == 0) *(u64 *)&d.colbits[p][(cbase + (tid & ~63)) >> 5] = mcol;
        // winners below each 32-column word of that bitmap = the word's first position in the winner list
        // (role_scan, TAB: the active word of an active column is act_list[rank of the column])
        if ((tid & 31) == 0) d.col_rank[p][c >> 5] = (uint16_t)(g_run + min(e_run, r));
    }
    if (c < d.sel_hi) {
        const bool sel = sel_any;
        if (own_col && (mode & EMIT_DUTY)) {
            float dc = my_duty * d.mom;
            if (sel) dc = dc + d.dinc;
            d.duty[c] = dc;
        }
        if (sel) {
            const int pos = (int)(g_run + min(e_run, r));
            (local ? d.cand_cols : d.active_cols[p])[pos] = c;
            s_col[pos - first_pos] = c;
            atomicAdd(&s_n, 1);
        }
        if (d.act[0] && (mode & EMIT_CLEAR)) {     // Temporal Memory present
            for (int h = 0; h < d.WPC; ++h) {
                d.pred[p][c * d.WPC + h] = 0;
                if (!sel || !(mode & EMIT_ACTIVATE)) { d.act[p][c * d.WPC + h] = 0; d.win[p][c * d.WPC + h] = 0; }
            }
        }
    }
    if (b == 0 && tid == 0 && !d.act[0]) d.ctr->step[p ^ 1] = d.ctr->step[p] + 1;   // SP-only handle
    EMIT_STAMP(5);                                   // (winner list and bitmap written)
    if (!tm_here && !local) return;
    __syncthreads();
    const int n_sel = s_n;
    if (local) {                                   // this block's candidates into the exchange record
        double *r_boost = (double *)d.send;
        uint32_t *r_col = (uint32_t *)(d.send + (size_t)d.cand_cap * 8), *r_win = r_col + d.cand_cap, *r_unacc = r_win + d.cand_cap;
        // with the cell words each would have if it became active (networks.py:95-104: they only depend on the owner's previous
        // predictions, segment maxima and segment counts): one candidate per half-wave, four per half-wave and pass with all
        // their loads in flight together
        const int has_distal = d.ctr->has_distal;
        const uint32_t step = d.ctr->step[p];
        constexpr int CPP = 4;
        for (int i0 = 0; i0 < n_sel; i0 += 8 * CPP) {
            int a[CPP];
            bool ok[CPP];
            ColumnLoads l[CPP];
            double bo[CPP];
#pragma unroll
            for (int u = 0; u < CPP; ++u) {
                const int i = i0 + u * 8 + (tid >> 5);
                ok[u] = i < n_sel;
                a[u] = ok[u] ? s_col[i] : d.c0;
                l[u] = tm_column_loads(d, p, ok[u], a[u], has_distal);
                bo[u] = d.boosted[p][a[u]];
            }
#pragma unroll
            for (int u = 0; u < CPP; ++u) {
                const ColumnWords w = tm_column_compute(d, 1, ok[u], a[u], ok[u] ? s_predw[a[u] - cbase] : 0u, l[u], has_distal, step);
                const int pos = first_pos + i0 + u * 8 + (tid >> 5);
                if (ok[u] && (lane & 31) == 0) {
                    r_boost[pos] = bo[u];
                    r_col[pos] = (uint32_t)a[u] | (w.burst ? 0x80000000u : 0u);
                    r_win[pos] = w.winner;
                    r_unacc[pos] = w.unacc;
                }
            }
        }
        for (int i = d.sel_k + b * 256 + tid; i < d.cand_cap; i += nblk * 256) ((u64 *)r_boost)[i] = CAND_PAD;      // (free slots)
        if (b == 0) {                              // and the segments that died while the previous step learned
            uint32_t *r_dead = r_unacc + d.cand_cap;
            const int n = min(d.dead_list[0], DEAD_CAP);
            if (tid == 0) {
                r_dead[0] = (uint32_t)n;
                uint32_t *r_n = (uint32_t *)(d.send + shard_record_count_offset(d.cand_cap));
                r_n[0] = (uint32_t)d.sel_k;         // (this path cuts exactly)
                r_n[1] = CAND_HOT_NONE;             // (... and builds no hot list: the global select reads all candidates)
                if (fused && wmode) d.ctr->cand_exact += 1;
            }
            for (int j = tid; j < n; j += 256) r_dead[1 + j] = (uint32_t)d.dead_list[1 + j];
        }
        return;
    }
    for (int i0 = 0; i0 < n_sel; i0 += tm_groups_per_block(d)) {       // a lane group per column
        const int i = i0 + tm_group_of(d, tid);
        const bool ok = i < n_sel;
        const int a = ok ? s_col[i] : 0;
        // (32 cell slots: the previous prediction words of the block's columns were staged with its keys)
        tm_activate_column(d, p, want_winner, ok, a, first_pos + i, d.WPC == 1 ? (u64)(ok ? s_predw[a - cbase] : 0u) : tm_pred_words(d, p, ok, a));
    }
}

// blocks [0, n_emit_blocks): the role; further blocks (a shard's candidates launch): zero the step's dense per-column words and
// the column bitmap (the winners' words are written after the exchange)
__global__ __launch_bounds__(256) void k_sp_emit(Dev d, int p, int want_winner, int fused, int mode, int wmode, int n_emit_blocks) {
    __shared__ EmitShared sh;
    if ((int)blockIdx.x >= n_emit_blocks) {
        const int nb = (int)gridDim.x - n_emit_blocks;
        for (int c = ((int)blockIdx.x - n_emit_blocks) * 256 + (int)threadIdx.x; c < d.C * d.WPC; c += nb * 256) {
            d.act[p][c] = 0;
            d.win[p][c] = 0;
            d.pred[p][c] = 0;
            if (c < d.colwords) d.colbits[p][c] = 0;
        }
        return;
    }
    role_emit(d, p, want_winner, fused, mode, blockIdx.x, n_emit_blocks, &sh, wmode);
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

// ---- SpatialPooler.process phase by phase (htm_sp_phase): kernels for values that come from the host ----------
// keys + top-digit histogram from boosted overlaps that are already in d.boosted[p] (a foreign boosting object computed
// them), or from overlaps in d.overlap[p] through the device's own boosting (a foreign proximal projection computed
// those): what role_overlap does after its popcounts
__global__ __launch_bounds__(RB) void k_sp_keys(Dev d, int p, int from_overlap) {
    __shared__ uint32_t h[SEL_BINS];
    const int gtid = blockIdx.x * RB + threadIdx.x, nthreads = gridDim.x * RB;
    uint32_t *ghist = d.hist + p * SEL_MAX_PASSES * SEL_BINS;
    if (gtid == 0) {
        d.ctr->emit_epoch += 1;
        d.ctr->sel_pass_prefix[p][0] = 0;
        d.ctr->sel_pass_krem[p][0] = (uint32_t)d.sel_k;
    }
    for (int i = gtid; i < (d.sel_passes - 1) * SEL_BINS; i += nthreads) ghist[SEL_BINS + i] = 0;
    for (int i = threadIdx.x; i < SEL_BINS; i += RB) h[i] = 0;
    __syncthreads();
    for (int c = d.sel_lo + gtid; c < d.sel_hi; c += nthreads) {
        double bo;
        if (from_overlap) {
            const float f = htm_exp_f32(d.coef * d.duty[c]);                 // regularizations.py:16
            bo = (double)f * (double)d.overlap[p][c];                        // :17
            d.boosted[p][c] = bo;
        } else {
            bo = d.boosted[p][c];
        }
        const u64 key = select_key(bo);
        d.key[p][c] = key;
        atomicAdd(&h[(uint32_t)(key >> sel_shift(0))], 1u);
    }
    __syncthreads();
    uint32_t *g0 = d.hist0 + (size_t)p * HIST0_PAR + (size_t)(blockIdx.x & (HIST_REP - 1)) * SEL_BINS;
    for (int i = threadIdx.x; i < SEL_BINS; i += RB)
        if (h[i]) atomicAdd(&g0[i], h[i]);
}

// a winner list that came from the host (already in d.active_cols[p], ascending): its column bitmap
__global__ __launch_bounds__(256) void k_sp_list_bits(Dev d, int p, int n, int pass) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (pass == 0) { if (i < d.colwords) d.colbits[p][i] = 0; return; }
    if (i < n) { const int c = d.active_cols[p][i]; atomicOr(&d.colbits[p][c >> 5], 1u << (c & 31)); }
}

// ExponentialBoosting.update (regularizations.py:19-21) as two launches: pass 0 `duty *= momentum` on every column,
// pass 1 `duty[active] += 1 - momentum` on the n listed ones (float32, two separately rounded operations)
__global__ __launch_bounds__(256) void k_sp_duty_list(Dev d, int p, int n, int pass) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (pass == 0) { if (i < d.C) d.duty[i] = d.duty[i] * d.mom; return; }
    if (i < n) { const int c = d.active_cols[p][i]; d.duty[c] = d.duty[c] + d.dinc; }
}

// a Spatial Pooler without Temporal Memory stepped phase by phase: close the step
__global__ void k_sp_commit(Dev d, int p) { d.ctr->step[p ^ 1] = d.ctr->step[p] + 1; }

// ---- column sharding: the kernels on either side of the exchange -----------------------------
// before the exchange: overlap + boost + histogram of the OWN columns (front: for the COMING step -- parity p ^ 1, the bank
// row after this step's -- where it does not ride in the previous step's last launch, k_learn_scan_overlap)
__global__ __launch_bounds__(RB) void k_shard_overlap(Dev d, const uint32_t *__restrict__ bank, int n_inputs, int G, int p, int wmode, int front) {
    __shared__ uint32_t h[SEL_BINS];
    role_overlap<RB>(d, bank, n_inputs, G, p, front ? p ^ 1 : p, front ? 1 : 0, blockIdx.x, gridDim.x, h, wmode);
}

// after the exchange: the exact global top-k over the world x KL candidates, computed by every block for itself
// (a radix select in LDS: the candidate keys are few and L2-resident), then block b emits rank b's winners into the
// ascending winner list -- the records are in rank order and each is in ascending column order, so the candidates
// ARE in ascending column order and the tie rule (lower column first) is their order -- together with the cell
// words the owner computed.  Block 0 also applies every rank's death reports to the replicated dead bits.
// The k-th largest of the keys a 1024-thread block holds in registers (bit j of vmask: kreg[j] is a key), and how many
// of the keys equal to it are among the k largest.  First the windowed pass of the three-launch schedule (win_bin): one
// histogram around `base` (the previous step's k-th key) -- 12-bit digits put thousands of similar keys into a handful
// of bins, and same-address LDS atomics run one after the other --, counted tie-aware; a chosen bin of up to 1 024 keys
// is ranked directly, a more crowded one (ties) finished by 12-bit digit passes from the window's resolution on.  A k-th key
// outside the window leaves it all to the digit passes, which start below the keys' common prefix (same result;
// *missed says so).  All threads call; h = SEL_BINS words of LDS.
struct BlockSelLds { uint32_t *h, *s_wave, *s_out, *s_cnt; u64 *s_or, *s_and; unsigned long long *trace; };
#ifdef BITHTM_SHARD_STAMPS                       // diagnostic build: device clock at the phases of block 0, d.trace[phase]
#define SHARD_STAMP(i) do { if (d.trace && blockIdx.x == 0 && threadIdx.x == 0) d.trace[i] = wall_clock64(); } while (0)
#define SEL_STAMP(i) do { if (L.trace && blockIdx.x == 0 && threadIdx.x == 0) L.trace[i] = wall_clock64(); } while (0)
#else
#define SHARD_STAMP(i) do { } while (0)
#define SEL_STAMP(i) do { } while (0)
#endif

template <int KPT>
__device__ __forceinline__ void block_select_regs(const u64 (&kreg)[KPT], uint32_t vmask, uint32_t k, uint32_t base, int low_zero,
                                                  const BlockSelLds &L, u64 *T_out, uint32_t *r_out, bool *missed, bool h_zeroed = false) {
    const int tid = threadIdx.x, lane = lane_id();
    uint32_t *h = L.h;
    u64 P = 0;
    uint32_t krem = k;
    bool done = false;
    u64 T = 0;
    int top_start = 64, top0 = 64;
    bool outside = false;
    {
        if (!h_zeroed) {                           // (the caller may have done this while its keys were on their way)
            for (int i = tid; i < SEL_BINS; i += 1024) h[i] = 0;
            if (tid == 0) { L.s_cnt[0] = 0; L.s_out[0] = 0; *L.s_or = 0; *L.s_and = ~0ull; }
            lds_barrier();
        }
        SEL_STAMP(8);
#pragma unroll
        for (int j = 0; j < KPT; ++j) {              // (keys below the window are not counted: the pick counts from the top and, if
            const uint32_t bin = win_bin(kreg[j], base);     // it gets that far, the k-th key is outside the window -- most keys of a
            hist_add_tie(h, bin, ((vmask >> j) & 1u) && bin != 0u);      // sharded step's candidates are: same-CU LDS atomics are what this pass costs)
        }
        lds_barrier();
        SEL_STAMP(9);
        uint32_t bucket, above;
        sel_pick<1024, true>(h, WIN_BINS, krem, L.s_wave, L.s_out, &bucket, &above);
        lds_barrier();
        SEL_STAMP(10);
        const bool inside = bucket >= 1u && bucket <= (WIN_COARSE << WIN_FINE);
        const uint32_t in_bin = inside ? h[bucket] : 0u;
        lds_barrier();
        if (inside && in_bin > 1024u) {                // a crowded bin: the digit passes below, from the window's resolution on
            const uint32_t fine = bucket - 1u;
            P = ((u64)(base + (fine >> WIN_FINE)) << 52) | ((u64)(fine & ((1u << WIN_FINE) - 1u)) << WIN_LOWBITS);
            krem -= above;
            top_start = WIN_LOWBITS;
        } else if (inside) {
            const uint32_t kb = krem - above;          // the kb-th largest of the bin's keys is the k-th overall
            u64 *list = (u64 *)h;                      // (the histogram is done with)
#pragma unroll
            for (int j = 0; j < KPT; ++j) {            // (one reservation per wave and pass: same-address LDS atomics run one after the other)
                const bool hit = ((vmask >> j) & 1u) && win_bin(kreg[j], base) == bucket;
                const u64 mh = __ballot(hit);
                if (!mh) continue;
                const int leader = __ffsll((long long)mh) - 1;
                uint32_t at = 0;
                if (lane == leader) at = atomicAdd(&L.s_cnt[0], (uint32_t)__popcll(mh));
                at = wave_read(at, leader);
                if (hit) list[at + __popcll(mh & lanemask_lt())] = kreg[j];
            }
            lds_barrier();
            for (uint32_t e = tid; e < in_bin; e += 1024) {
                const u64 ke = list[e];
                uint32_t ng = 0, nq = 0;
                for (uint32_t f = 0; f < in_bin; ++f) {
                    const u64 kf = list[f];
                    ng += kf > ke;
                    nq += kf == ke;
                }
                if (ng < kb && kb <= ng + nq) { *L.s_or = ke; L.s_out[1] = kb - ng; }      // (equal keys write the same pair)
            }
            lds_barrier();
            T = *L.s_or;
            krem = L.s_out[1];
            done = true;
            lds_barrier();
        } else {
            // the k-th key is outside the window: everything is left to the digit passes, which start below the keys' common
            // prefix (the bits that are the same in every key; only computed here, where they are needed)
            u64 vo = 0, va = ~0ull;
#pragma unroll
            for (int j = 0; j < KPT; ++j)
                if ((vmask >> j) & 1u) { vo |= kreg[j]; va &= kreg[j]; }
            vo = wave_reduce64(vo, 0ull, [](u64 a, u64 b) { return a | b; });
            va = wave_reduce64(va, ~0ull, [](u64 a, u64 b) { return a & b; });
            if (lane == 0) { atomicOr((unsigned long long *)L.s_or, vo); atomicAnd((unsigned long long *)L.s_and, va); }
            lds_barrier();
            const u64 differ = *L.s_or ^ *L.s_and;
            top0 = differ ? 64 - __clzll((long long)differ) : 0;       // bits [top0, 64) are the same in every key
            P = top0 < 64 ? (*L.s_and >> top0) << top0 : 0ull;
            top_start = top0;
            outside = true;
            lds_barrier();
        }
    }
    for (int top = top_start; !done && top > low_zero;) {
        const int bits = min(SEL_DIGIT, top - low_zero), shift = top - bits, nb = 1 << bits;
        for (int i = tid; i < nb; i += 1024) h[i] = 0;
        lds_barrier();
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const u64 kk = kreg[j];
            hist_add_tie(h, (uint32_t)(kk >> shift) & (nb - 1), ((vmask >> j) & 1u) && (top >= 64 || ((kk ^ P) >> top) == 0));
        }
        lds_barrier();
        uint32_t bucket, above;
        sel_pick<1024, true>(h, nb, krem, L.s_wave, L.s_out, &bucket, &above);
        P |= (u64)bucket << shift;
        krem -= above;
        top = shift;
        lds_barrier();
    }
    *T_out = done ? T : P;
    *r_out = krem;
    *missed = outside;
}

__global__ __launch_bounds__(1024) void k_shard_select(Dev d, const unsigned char *__restrict__ recv, int p) {
    __shared__ uint32_t h[SEL_BINS];
    __shared__ uint32_t s_wave[16], s_out[2], s_cnt[2], s_rank[2];
    __shared__ u64 s_or, s_and;
    const int tid = threadIdx.x, lane = lane_id();
    const int KL = d.cand_cap, n_tot = d.world * KL;          // candidate SLOTS; rank r filled the first count_of(r) of its own
    const size_t rb = shard_record_bytes(KL), cnt_off = shard_record_count_offset(KL);
    if ((int)blockIdx.x == d.world) {
        // one block beside the select (it used to be the tail of block 0, 2-3 us): every rank's death reports, all at once
        // (rank after rank, each with its dependent loads and atomics, this was 8 x 2 round trips): thread -> (rank, entry)
        // through the counts
        __shared__ int s_dead[64 + 1];
        if (tid <= 64) s_dead[tid] = 0;
        __syncthreads();
        const int nr = min(d.world, 64);
        if (tid < nr) s_dead[tid + 1] = min((int)((const uint32_t *)(recv + (size_t)tid * rb + (size_t)KL * 20))[0], DEAD_CAP);
        __syncthreads();
        if (tid == 0) for (int r = 0; r < nr; ++r) s_dead[r + 1] += s_dead[r];
        __syncthreads();
        const int total_dead = s_dead[nr];
        for (int e = tid; e < total_dead; e += 1024) {
            int r = 0;
            while (s_dead[r + 1] <= e) ++r;
            const uint32_t *r_dead = (const uint32_t *)(recv + (size_t)r * rb + (size_t)KL * 20);
            const int gid = (int)r_dead[1 + e - s_dead[r]];
            const uint32_t old = atomicOr(&d.dead_bits[gid >> 5], 1u << (gid & 31));
            if (!((old >> (gid & 31)) & 1u)) recyc_add(d, gid >> 10, 1);
        }
        if (tid == 0) d.dead_list[0] = 0;          // reported; the coming learning role collects this step's
        return;
    }
    auto emit_winner = [&](int pos, uint32_t cw, uint32_t wn, uint32_t un) {
        const int col = (int)(cw & 0x7FFFFFFFu);
        const bool burst = cw >> 31;
        const uint32_t act = burst ? cell_mask(d.K) : wn;                       // networks.py:115
        d.active_cols[p][pos] = col;
        atomicOr(&d.colbits[p][col >> 5], 1u << (col & 31));
        d.act[p][col] = act;
        d.win[p][col] = wn;
        d.bursting[pos] = burst ? 1 : 0;
        d.actw_id[pos] = col;                       // (a sharded handle has one cell word per column)
        d.unacc_word[pos] = un;
        d.winw_idx[pos] = wn;
        d.actcnt[pos] = (uint8_t)__popc(act);
    };
    SHARD_STAMP(0);
    {
        // ---- the short way: the ranks' hot lists (see the record's layout).  Every rank's list within the budget, k keys
        // between them and their k-th largest at or above every list's floor: it is the k-th largest of all candidates, and
        // the winners are hot.  The block reads world x budget keys, its own rank's hot slots and, through them, the words of
        // its own hot candidates -- a few KB -- and nothing of the candidates' arrays.
        constexpr int KH = SHARD_HOT_KEYS;         // hot keys per thread
        __shared__ uint32_t s_fast[2];
        __shared__ u64 s_floor;
        const int budget = d.hot_budget;
        const size_t hot_off = shard_record_hot_offset(KL);
        const uint32_t win_base = d.ctr->sel_win_global;
        const int bq = blockIdx.x;
        const unsigned char *rec = recv + (size_t)bq * rb;
        const uint32_t *hdr = (const uint32_t *)(recv + (size_t)min(tid, d.world - 1) * rb + cnt_off);     // (threads < world use it)
        const uint32_t nh = hdr[1];
        const u64 floor_r = ((u64)hdr[3] << 32) | hdr[2];
        const uint32_t n_own_raw = ((const uint32_t *)(rec + cnt_off))[1];
        u64 hk[KH];
        bool in_list[KH];
#pragma unroll
        for (int j = 0; j < KH; ++j) {
            const int i = tid + j * 1024, r = i / budget, e = i - r * budget;
            in_list[j] = r < d.world;
            hk[j] = ((const u64 *)(recv + (size_t)min(r, d.world - 1) * rb + hot_off))[e];
        }
        u64 okey[2];
        uint32_t oslot[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = min(2 * tid + u, KL - 1);
            okey[u] = ((const u64 *)(rec + hot_off))[e];
            oslot[u] = ((const uint16_t *)(rec + hot_off + (size_t)KL * 8))[e];
        }
        // while they travel: the histogram of the windowed pass, zeroed (its barrier does not wait for the loads)
        for (int i = tid; i < SEL_BINS; i += 1024) h[i] = 0;
        if (tid == 0) { s_cnt[0] = 0; s_out[0] = 0; s_or = 0; s_and = ~0ull; s_rank[0] = 0; s_rank[1] = 0; s_fast[0] = 0; s_fast[1] = 0; s_floor = 0; }
        lds_barrier();
        if (tid < d.world) {
            if (nh == CAND_HOT_NONE || nh > (uint32_t)budget) atomicOr(&s_fast[0], 1u);
            else { atomicAdd(&s_fast[1], nh); atomicMax((unsigned long long *)&s_floor, floor_r); }
        }
        // the own hot candidates' words, asked for as soon as their slots are here
        const uint32_t *r_col = (const uint32_t *)(rec + (size_t)KL * 8), *r_win = r_col + KL, *r_unacc = r_win + KL;
        uint32_t cw[2], wn[2], un[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int slot = min((int)oslot[u], KL - 1);
            cw[u] = r_col[slot];
            wn[u] = r_win[slot];
            un[u] = r_unacc[slot];
        }
        lds_barrier();
        if (s_fast[0] == 0 && s_fast[1] >= (uint32_t)d.k) {
            const int n_own = (int)n_own_raw;          // (<= budget <= 2 048: one pass of two entries per thread)
            uint32_t vmask = 0;
#pragma unroll
            for (int j = 0; j < KH; ++j) {
                const bool ok = in_list[j] && hk[j] != CAND_PAD;
                hk[j] = ok ? select_key_bits(hk[j]) : 0ull;
                vmask |= (ok ? 1u : 0u) << j;
            }
            u64 T;
            uint32_t krem;
            bool missed = false;
            const BlockSelLds L{h, s_wave, s_out, s_cnt, &s_or, &s_and, d.trace};
            SHARD_STAMP(1);
            block_select_regs<KH>(hk, vmask, (uint32_t)d.k, win_base, d.low_zero, L, &T, &krem, &missed, true);
            SHARD_STAMP(2);
            if (T >= s_floor) {                        // (else: a rank may hold a better candidate outside its list -- the long way)
            {   // winners among the hot candidates of the ranks before this one
                uint32_t g = 0, e = 0;
#pragma unroll
                for (int j = 0; j < KH; ++j) {
                    const bool before = tid + j * 1024 < bq * budget && ((vmask >> j) & 1u);
                    g += before && hk[j] > T;
                    e += before && hk[j] == T;
                }
                g = wave_sum(g);
                e = wave_sum(e);
                if (lane == 0 && (g | e)) { atomicAdd(&s_rank[0], g); atomicAdd(&s_rank[1], e); }
            }
            uint32_t flag[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const u64 kk = select_key_bits(okey[u]);
                flag[u] = 2 * tid + u < n_own ? ((kk > T) ? 1u : ((kk == T) ? 0x10000u : 0u)) : 0u;
            }
            uint32_t total;
            uint32_t ex = block_excl_scan<1024, true>(flag[0] + flag[1], s_wave, total);
            const uint32_t gt_run = s_rank[0], eq_run = s_rank[1];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t g = gt_run + (ex & 0xFFFFu), e = eq_run + (ex >> 16);
                if ((flag[u] & 1u) || ((flag[u] >> 16) && e < krem)) emit_winner((int)(g + min(e, krem)), cw[u], wn[u], un[u]);
                ex += flag[u];
            }
            SHARD_STAMP(3);
            if (bq == 0 && tid == 0) {
                d.ctr->sel_win_global = min(win_base_global(T) + (uint32_t)d.win_offset, 4096u - WIN_COARSE);
                if (missed) d.ctr->sel_fallbacks += 1;
                d.ctr->hot_selects += 1;
            }
            SHARD_STAMP(5);
            return;
            }
#ifdef BITHTM_SHARD_STAMPS
            if (d.trace && bq == 0 && tid == 0) atomicAdd(&d.trace[20], 1ull);
#endif
            lds_barrier();                             // (everybody has read T's LDS words: the long way starts over)
        }
#ifdef BITHTM_SHARD_STAMPS                       // (why not: a rank without a hot list / over the budget / fewer than k hot keys)
        if (d.trace && bq == 0 && tid < d.world) {
            if (nh == CAND_HOT_NONE) atomicAdd(&d.trace[16], 1ull);
            else if (nh > (uint32_t)budget) atomicAdd(&d.trace[17], 1ull);
        }
        if (d.trace && bq == 0 && tid == 0) { if (s_fast[0] == 0) atomicAdd(&d.trace[18], 1ull); d.trace[19] = s_fast[1]; }
#endif
    }
    // ---- the long way: every candidate of every rank
    auto count_of = [&](int r) -> int { return min((int)*(const uint32_t *)(recv + (size_t)r * rb + cnt_off), KL); };
    const u64 inv_kl = ((1ull << 32) + (u64)KL - 1) / (u64)KL;      // i / KL for i < 2^16-ish: one multiplication (64 bits: KL = 1 gives 2^32)
    auto key_at = [&](int i, bool *filled) -> u64 {      // the key in slot i and whether the slot holds a candidate (CAND_PAD: free)
        int r = (int)(((u64)(uint32_t)i * inv_kl) >> 32);
        if (r * KL > i) --r;
        const int j = i - r * KL;
        const u64 bits = ((const u64 *)(recv + (size_t)r * rb))[j];
        *filled = bits != CAND_PAD;
        return select_key_bits(bits);
    };
    // up to KPT keys per thread stay in registers over the passes (configs[3] 8-way: 13 104 slots, 13 per thread);
    // beyond that they are read again, from L2
    constexpr int KPT = 13;
    const bool in_regs = n_tot <= KPT * 1024;
    // everything this block will need that does not depend on the k-th key is asked for here, all at once: the window's base,
    // the own rank's count and the words of its first 2 048 slots (own winners, below)
    const uint32_t win_base = d.ctr->sel_win_global;
    const int n_own = count_of((int)blockIdx.x);
    const unsigned char *rec = recv + (size_t)blockIdx.x * rb;
    const uint32_t *r_col = (const uint32_t *)(rec + (size_t)KL * 8), *r_win = r_col + KL, *r_unacc = r_win + KL;
    u64 own_bits[2];
    uint32_t own_cw[2], own_wn[2], own_un[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = min(2 * tid + u, KL - 1);
        own_bits[u] = ((const u64 *)rec)[j];
        own_cw[u] = r_col[j];
        own_wn[u] = r_win[j];
        own_un[u] = r_unacc[j];
    }
    u64 kreg[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {                // (clamped and unconditional: a branch around a load makes the compiler wait for it)
        const int i = min(tid + j * 1024, n_tot - 1);
        int r = (int)(((u64)(uint32_t)i * inv_kl) >> 32);
        if (r * KL > i) --r;
        kreg[j] = ((const u64 *)(recv + (size_t)r * rb))[i - r * KL];          // (raw bits for now)
    }
    // while they travel: the histogram of the windowed pass, zeroed (its barrier does not wait for the loads)
    for (int i = tid; i < SEL_BINS; i += 1024) h[i] = 0;
    if (tid == 0) { s_cnt[0] = 0; s_out[0] = 0; s_or = 0; s_and = ~0ull; s_rank[0] = 0; s_rank[1] = 0; }
    lds_barrier();
    uint32_t vmask = 0;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const bool ok = in_regs && tid + j * 1024 < n_tot && kreg[j] != CAND_PAD;
        kreg[j] = ok ? select_key_bits(kreg[j]) : 0ull;
        vmask |= (ok ? 1u : 0u) << j;
    }
    u64 T;                                         // the k-th largest key; krem of the keys equal to it win
    uint32_t krem;
    bool missed = false;
    if (in_regs) {
        const BlockSelLds L{h, s_wave, s_out, s_cnt, &s_or, &s_and, d.trace};
        SHARD_STAMP(1);                            // (keys loaded)
        block_select_regs<KPT>(kreg, vmask, (uint32_t)d.k, win_base, d.low_zero, L, &T, &krem, &missed, true);
    } else {
        // more candidates than the registers hold (configs[4]: 42 k): 12-bit digit passes over the records, from below the
        // keys' common prefix
        if (tid == 0) { s_or = 0; s_and = ~0ull; }
        __syncthreads();
        {
            u64 vo = 0, va = ~0ull;
            for (int i = tid; i < n_tot; i += 1024) {
                bool filled;
                const u64 kk = key_at(i, &filled);
                if (filled) { vo |= kk; va &= kk; }
            }
            vo = wave_reduce64(vo, 0ull, [](u64 a, u64 b) { return a | b; });
            va = wave_reduce64(va, ~0ull, [](u64 a, u64 b) { return a & b; });
            if (lane == 0) { atomicOr((unsigned long long *)&s_or, vo); atomicAnd((unsigned long long *)&s_and, va); }
        }
        __syncthreads();
        SHARD_STAMP(1);
        const u64 differ = s_or ^ s_and;
        const int top0 = differ ? 64 - __clzll((long long)differ) : 0;
        u64 P = top0 < 64 ? (s_and >> top0) << top0 : 0ull;
        krem = (uint32_t)d.k;
        for (int top = top0; top > d.low_zero;) {
            const int bits = min(SEL_DIGIT, top - d.low_zero), shift = top - bits, nb = 1 << bits;
            for (int i = tid; i < nb; i += 1024) h[i] = 0;
            __syncthreads();
            for (int i0 = 0; i0 < n_tot; i0 += 1024) {        // (whole waves: hist_add_tie is called by all lanes)
                const int i = i0 + tid;
                bool filled;
                const u64 kk = key_at(min(i, n_tot - 1), &filled);
                hist_add_tie(h, (uint32_t)(kk >> shift) & (nb - 1), i < n_tot && filled && (top >= 64 || ((kk ^ P) >> top) == 0));
            }
            __syncthreads();
            uint32_t bucket, above;
            sel_pick<1024>(h, nb, krem, s_wave, s_out, &bucket, &above);
            P |= (u64)bucket << shift;
            krem -= above;
            top = shift;
            __syncthreads();
        }
        T = P;
    }
    SHARD_STAMP(2);                                // (k-th key known)
    const int b = blockIdx.x, lo = b * KL;
    {   // winners among the candidates of the ranks before this one (s_rank was zeroed at the top; the sums are read behind the
        // barriers of the scan below)
        uint32_t g = 0, e = 0;
        if (in_regs) {
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const bool before = tid + j * 1024 < lo && ((vmask >> j) & 1u);
                g += before && kreg[j] > T;
                e += before && kreg[j] == T;
            }
        } else {
            for (int i = tid; i < lo; i += 1024) {
                bool filled;
                const u64 kk = key_at(i, &filled);
                g += filled && kk > T;
                e += filled && kk == T;
            }
        }
        g = wave_sum(g);
        e = wave_sum(e);
        if (lane == 0 && (g | e)) { atomicAdd(&s_rank[0], g); atomicAdd(&s_rank[1], e); }
    }
    // the own rank's winners, two neighbouring slots per thread and pass (one scan for 2 048 slots; the first pass's words
    // have been here since the top of the kernel)
    uint32_t gt_run = 0, eq_run = 0;
    for (int j0 = 0; j0 < n_own; j0 += 2048) {
        u64 bits[2];
        uint32_t cw[2], wn[2], un[2], flag[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = j0 + 2 * tid + u;
            if (j0 == 0) { bits[u] = own_bits[u]; cw[u] = own_cw[u]; wn[u] = own_wn[u]; un[u] = own_un[u]; }
            else {
                const int jc = min(j, KL - 1);
                bits[u] = ((const u64 *)rec)[jc]; cw[u] = r_col[jc]; wn[u] = r_win[jc]; un[u] = r_unacc[jc];
            }
            const u64 kk = select_key_bits(bits[u]);
            flag[u] = j < n_own ? ((kk > T) ? 1u : ((kk == T) ? 0x10000u : 0u)) : 0u;
        }
        uint32_t total;
        uint32_t ex = block_excl_scan<1024, true>(flag[0] + flag[1], s_wave, total);      // (LDS-only barriers: a later pass does not
        if (j0 == 0) { gt_run = s_rank[0]; eq_run = s_rank[1]; }                          // wait for an earlier one's stores)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t g = gt_run + (ex & 0xFFFFu), e = eq_run + (ex >> 16);
            if ((flag[u] & 1u) || ((flag[u] >> 16) && e < krem)) emit_winner((int)(g + min(e, krem)), cw[u], wn[u], un[u]);
            ex += flag[u];
        }
        gt_run += total & 0xFFFFu;
        eq_run += total >> 16;
    }
    SHARD_STAMP(3);
    SHARD_STAMP(4);                                // (own winners emitted)
    if (b == 0 && tid == 0) {                      // (last: a store ahead of a barrier is waited for)
        d.ctr->sel_win_global = min(win_base_global(T) + (uint32_t)d.win_offset, 4096u - WIN_COARSE);
        if (missed) d.ctr->sel_fallbacks += 1;     // (telemetry: the window missed)
    }
    SHARD_STAMP(5);
}

#endif
