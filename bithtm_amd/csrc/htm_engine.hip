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
// Internal cell encoding: enc = column * KP + cell, KP = 32 cell slots per column (one 32-bit word per column in the dense
// bitmaps) or 64 for cell_dim above 32 (two words);
// the ABI converts to / from the reference's flat id column * cell_dim + cell.

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

#include "../../include/bithtm_hip.h"
#include "htm_fexp.h"
#include "htm_rng.h"

#include "htm_dev.h"
#include "htm_sp_kernels.h"
#include "htm_tm_kernels.h"
#include "htm_pipeline.h"

// ------------------------------------------------------------------------------------------
// host side

struct ncclUniqueIdBytes { char internal[128]; };      // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES), passed by value

static thread_local std::string g_create_error;
static int (*g_rccl_destroy)(void *) = nullptr;        // ncclCommDestroy once RCCL is loaded

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
    bool shard_graph_ok;                  // htm_shard_comm_init's preflight: this communicator's all-gather replays correctly from a captured hipGraph
    int shard_front_wmode;                // htm_shard_run: the histogram form of the overlap computed ahead for the coming step
    // keyed by the exact launch parameters: (parity | learning | front flags | span, ranks, n_inputs, per rank: scan blocks known to have
    // segments, scan form, exchange mode) and the bank -- (on rank 0's handle for a group in one process)
    std::map<std::pair<std::vector<int>, const void *>, hipGraphExec_t> shard_graphs;
    void *rccl_comm;                      // ncclComm_t of htm_shard_comm_init (the exchange of htm_shard_step)
    unsigned char *shard_send, *shard_recv;   // ... and its device buffers
    int G;                                // lanes per SP row
    int graph_steps;                      // steady-state steps per captured graph (BITHTM_GRAPH_STEPS)
    int eager_below;                      // calls of fewer steps than this launch eagerly (BITHTM_EAGER_BELOW)
    bool emit_fused, emit_fused_open;     // the emit grid is resident at once in k_sp_emit / in k_open_emit (refreshed per call)
    bool emit_fits_lean;                  // ... and in k_learn_scan_emit
    int knob_lean, knob_fuse_tm, knob_shard_window, knob_scan_large, knob_step_window, knob_tail_rows, knob_step_split;
    int scan_large_above;                 // segments above which the scan takes its streaming (large-pool) form (BITHTM_SCAN_LARGE_ABOVE)
    int knob_scan_dyn;                    // the large-pool form of the last launch: every block joins the streaming scan when its own role is done (BITHTM_SCAN_DYN)
    int lean_resident_large;              // blocks of the large-pool k_learn_scan_emit that are resident at once
    int knob_defer_tail;                  // htm_step holds a step's last launch back for the next call's first (BITHTM_DEFER_TAIL)
    bool tail_pending;                    // ... and one is held back now: the learning role and the scan of the step of parity tail_p
    int tail_p;
    bool window_known;                    // a select has run on this handle since it was created / imported into: Counters::sel_win is meaningful      // environment knobs, read when the handle is created
    bool emit_fits, emit_fits_open;       // ... as far as this handle's own grids go (fixed at creation)
    int sel_passes_fused, sel_passes_full; // launched select digits with / without the in-kernel finish
    int seg_hint;                         // a lower bound of the segment count (see scan_spec_blocks)
    int *seg_pinned;                      // pinned word the end of each htm_run copies the count into
    int sp_blocks, sel_blocks, c256_blocks, s1024_blocks, scan_blocks, zero_blocks, lean_learn_blocks, lean_learn_blocks_large, lean_scan_blocks, lean_scan_blocks_large, lean_overlap_blocks, lean2_classify_blocks, lean2_order, cus;
    bool lean2_classify_set;               // BITHTM_LEAN2_CLASSIFY given: that many classification blocks whatever the pool's size
    const uint32_t *ahead_bank;           // htm_run ended with HTM_RUN_CONTINUE on this bank: the SP has done the next step
    int ahead_n_inputs, ahead_learning;   //   and the front of the one after it
    bool ahead_lean;                      //   ... in the three-launch schedule (else the four-launch one)
    int phase_active;                     // htm_sp_phase: length of the current winner list
    bool import_keep;                     // htm_import_begin(HTM_IMPORT_PREV_STATE): the commit leaves the store, the step index and the sticky flags alone
    bool phase_open;                      // ... phases of the current (not yet closed) timestep have run: the Spatial Pooler
                                          // fields htm_read returns are that step's
    // graphs keyed by (parity, learning, bank, n_inputs)
    std::map<std::tuple<int, int, const void *, int>, hipGraphExec_t> graphs;
    // state import staging (htm_write of the MATCH_* / SEG_POTENTIAL fields, applied at commit)
    std::vector<int> imp_pot, imp_match_seg;
    // ... on a column-sharded handle also the per-segment arrays (written for ALL segment ids; the commit keeps the rows of
    // the segments this rank's cells own)
    std::vector<int> imp_seg_cell, imp_seg_nsyn, imp_presyn;
    std::vector<float> imp_perm;
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

// Live handles, per process.  The in-kernel select finish (k_sp_emit / k_open_emit) makes the blocks of one grid
// wait for each other, which is only safe while nothing else can hold the CU slots that grid needs: kernels of
// one stream run one after the other, kernels of another handle's stream do not.  A handle therefore uses the
// in-kernel finish only while every other live handle on its device enqueues on the same stream; otherwise it
// launches all select digits and the separate count kernel (slower, no waiting between blocks, same result).
static std::mutex g_registry_mutex;
static std::vector<htm_handle *> g_registry;

static void refresh_exchange_mode(htm_handle *h) {
    bool solo = true;
    {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        for (const htm_handle *o : g_registry)
            if (o != h && o->device == h->device && o->stream != h->stream) solo = false;
    }
    h->emit_fused = h->emit_fits && solo;
    h->emit_fused_open = h->emit_fits_open && solo;
    h->d.sel_passes = h->emit_fused ? h->sel_passes_fused : h->sel_passes_full;
}

#define HIPCHK(h, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
            return HTM_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

// the Spatial Pooler is ahead of the Temporal Memory across a call boundary (htm_run with HTM_RUN_CONTINUE): only the
// continuing htm_run may come next
static bool sp_is_ahead(const htm_handle *h) { return h->ahead_bank != nullptr; }
#define REJECT_WHEN_AHEAD(h)                                                                                      \
    do {                                                                                                          \
        if (sp_is_ahead(h)) {                                                                                     \
            (h)->err = "the Spatial Pooler is ahead (htm_run ended with HTM_RUN_CONTINUE): continue with htm_run on the same bank"; \
            return HTM_ERR_STATE;                                                                                 \
        }                                                                                                         \
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

static size_t learn_lds(int epl, int bs = RB) { return (size_t)(bs / 64) * CAND_CAP * 8 + (size_t)(bs / 64) * epl * 64 * 4 + (size_t)WIN_LDS * 4; }
static int learn_epl(const Dev &d) { const int e = d.E / 64; return e <= 1 ? 1 : e == 2 ? 2 : e <= 4 ? 4 : 8; }
// (... and, behind the bitmap, the queues in which the waves of the streaming form set rows aside: 32 entries of 12 bytes per wave)
static size_t scan_lds(const Dev &d, int use_lds) { return 16 + (use_lds ? (size_t)((d.colwords + 3) & ~3) * 4 : 0) + 4 * 32 * 12; }
// ... with the rank of every bitmap word and the active words of the step's active columns behind the bitmap (the
// three-launch schedule's scan looks the cells of active columns up in LDS: role_scan, TAB)
static size_t scan_lds_tab(const Dev &d) {
    return 16 + (size_t)((d.colwords + 3) & ~3) * 4 + (size_t)((d.colwords * 2 + 15) / 16) * 16 + (size_t)((d.k * d.WPC + 8 + 3) / 4) * 16;
}
// ... which needs them to fit (and the ranks 16 bits); otherwise the schedule's scan reads the cell words from memory
static bool lean_tab(const Dev &d) { return scan_lds_tab(d) <= 64 * 1024 && d.k * d.WPC < 65536; }
static size_t lean_scan_lds(const Dev &d) { return lean_tab(d) ? scan_lds_tab(d) : scan_lds(d, 1); }
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
static bool scan_pool_is_large(const htm_handle *h) {
    if (h->knob_scan_large >= 0) return h->knob_scan_large != 0;      // (BITHTM_SCAN_LARGE: tuning knob)
    return h->seg_hint > h->scan_large_above;
}

static void launch_scan(htm_handle *h, int p, int use_lds) {
    Dev &d = h->d;
    const int spec = scan_spec_blocks(h);
    if (scan_pool_is_large(h) && use_lds && scan_lds(d, 1) > 16 * 1024) {
        // a big column bitmap (32 KB at 262 144 columns): one copy per 1024-thread block, two blocks per CU
        const int blocks = std::max(1, std::min((d.Lcap + 255) / 256, 2 * h->cus));
        LAUNCH_ON(h, h->stream, scan_lds(d, 1), "tm_scan_wide", (k_tm_scan_wide<true>), blocks, 1024, d, p, 0);
        return;
    }
    if (scan_pool_is_large(h)) {
        if (use_lds) LAUNCH_ON(h, h->stream, scan_lds(d, 1), "tm_scan_large", (k_tm_scan<true, 1>), h->scan_blocks, 256, d, p, spec);
        else LAUNCH_ON(h, h->stream, scan_lds(d, 0), "tm_scan_large", (k_tm_scan<false, 1>), h->scan_blocks, 256, d, p, spec);
    } else {
        if (use_lds) LAUNCH_ON(h, h->stream, scan_lds(d, 1), "tm_scan", (k_tm_scan<true, 6>), h->scan_blocks, 256, d, p, spec);
        else LAUNCH_ON(h, h->stream, scan_lds(d, 0), "tm_scan", (k_tm_scan<false, 6>), h->scan_blocks, 256, d, p, spec);
    }
}

// the windowed one-pass select (win_bin) outside the three-launch schedule too: the histogram is finished inside the emit
// grid, so only where that grid's blocks may wait for each other (BITHTM_STEP_WINDOW=0: two launched digits, as before)
// (and only once a select has left a window behind: the first step of a handle, or after a state import, takes the digits)
static int step_wmode(const htm_handle *h) { return h->emit_fused && h->knob_step_window && h->world == 1 && h->window_known ? 1 : 0; }

// Front of SpatialPooler.process for the step with parity sp: overlap + boost + histogram (the windowed one, or the top
// key digit and then the remaining select digits).  host_input: the packed input of a host-fed step (it rides in the
// launch's arguments where it fits); else the bank in device memory.
static void enqueue_sp_front(htm_handle *h, const uint32_t *bank, int n_inputs, int p, int wmode, const uint32_t *host_input = nullptr) {
    Dev &d = h->d;
    if (host_input && d.W <= ARG_INPUT_WORDS) {
        PackedInputArg in;
        memset(&in, 0, sizeof(in));
        memcpy(in.w, host_input, (size_t)((d.I + 31) / 32) * 4);
        LAUNCH(h, "sp_overlap", k_sp_overlap_arg, h->sp_blocks, RB, d, in, h->G, p, wmode);
    } else {
        LAUNCH(h, "sp_overlap", k_sp_overlap, h->sp_blocks, RB, d, bank, n_inputs, h->G, p, p, 0, wmode);
    }
    if (!wmode)
        for (int pass = 1; pass < d.sel_passes; ++pass) LAUNCH(h, "sp_select", k_sel_pass, h->sel_blocks, RB, d, pass, p);
}

// Rest of SpatialPooler.process: count + emit.  mode = EMIT_ALL: all of it, with the TM's per-column
// activation when the handle has a Temporal Memory.  sp_learn: the permanence update as a launch of
// its own (handles without a Temporal Memory, and the first step of a pipelined run).
static void enqueue_sp_back(htm_handle *h, const uint32_t *bank, int n_inputs, int p, int want_winner, int mode, bool sp_learn, int wmode = 0) {
    Dev &d = h->d;
    const int fused = h->emit_fused;               // all blocks co-resident: count inside emit
    if (!fused) LAUNCH(h, "sp_count", k_sp_count, h->c256_blocks, 256, d, p);
    LAUNCH(h, "sp_emit", k_sp_emit, h->c256_blocks, 256, d, p, want_winner, fused, mode, fused ? wmode : 0, h->c256_blocks);
    h->window_known = true;                         // (every select leaves the next step's window behind)
    // (a forked graph branch for this independent update was measured at +17..29 us per step on
    // this runtime, against 2.3 us for one more kernel in the chain: tools/launch_overhead.hip)
    if (sp_learn) LAUNCH(h, "sp_learn", k_sp_learn, d.k, 256, d, bank, n_inputs, p);
}

// TemporalMemory.process after the per-column activation, one role per launch.  sp_rows: the SP
// permanence update of this step rides along in the middle launch.
// front_wmode >= 0 (column-sharded handles inside htm_shard_run): the overlap of the COMING step on the rank's own
// columns rides in the last launch (wmode = front_wmode: windowed histogram or top digit)
// the learning role and the scan: one launch (the learning waves scan their own rows), unless the pool is large (the
// streaming scan kernels) or somebody is timing the roles one by one (unsharded handles under htm_profile)
static bool tm_tail_fused(const htm_handle *h) {
    return h->knob_fuse_tm && !(h->profile && h->world == 1) && !scan_pool_is_large(h) && scan_lds(h->d, 1) <= 64 * 1024;
}

// the learning role and the scan of the step of parity p, nothing beside them
static void enqueue_tm_tail(htm_handle *h, int p) {
    Dev &d = h->d;
    if (tm_tail_fused(h)) {
        const int epl = learn_epl(d), n_learn = h->lean_learn_blocks, n_scan = h->lean_scan_blocks, spec = scan_spec_blocks(h);
        const size_t lds = std::max(learn_lds(epl, 256), scan_lds(d, 1));
        switch (epl) {
            case 1: LAUNCH_ON(h, h->stream, lds, "tm_learn+tm_scan", (k_learn_scan_emit<1, 6>), n_learn + n_scan, 256, d, p, 0, n_learn, n_scan, spec); break;
            case 2: LAUNCH_ON(h, h->stream, lds, "tm_learn+tm_scan", (k_learn_scan_emit<2, 6>), n_learn + n_scan, 256, d, p, 0, n_learn, n_scan, spec); break;
            case 4: LAUNCH_ON(h, h->stream, lds, "tm_learn+tm_scan", (k_learn_scan_emit<4, 6>), n_learn + n_scan, 256, d, p, 0, n_learn, n_scan, spec); break;
            default: LAUNCH_ON(h, h->stream, lds, "tm_learn+tm_scan", (k_learn_scan_emit<8, 6>), n_learn + n_scan, 256, d, p, 0, n_learn, n_scan, spec); break;
        }
    } else {
        launch_learn(h, p);
        launch_scan(h, p, scan_lds(d, 1) <= 64 * 1024);
    }
}

// htm_step may hold a step's last launch back (the next call's first launch carries it beside its overlap): everything else
// that touches the handle lets it go first
static void flush_tail(htm_handle *h) {
    if (!h->tail_pending) return;
    h->tail_pending = false;
    hipSetDevice(h->device);
    enqueue_tm_tail(h, h->tail_p);
}

// defer_tail: the last launch is held back (htm_step: see flush_tail); the permanence rows then ride in the middle launch
static void enqueue_tm(htm_handle *h, int n_active, int learning, int want_winner, int p,
                       const uint32_t *bank, int n_inputs, bool sp_rows, int front_wmode = -1, bool defer_tail = false) {
    Dev &d = h->d;
    const int n_cls = learning ? kClassifyBlocks : 0;
    const bool fuse = tm_tail_fused(h);
    int n_sp_rows = (sp_rows && learning && h->cfg.enable_sp) ? d.k : 0;
    // an unsharded step's permanence rows ride in that launch (the middle launch is left with the Temporal Memory's chain);
    // a shard's stay in the middle launch: the coming step's overlap, which reads them, may ride in the last one
    const int n_tail_rows = (fuse && h->world == 1 && h->knob_tail_rows && !defer_tail) ? n_sp_rows : 0;
    if (n_tail_rows) n_sp_rows = 0;
    const int n_duty = h->world > 1 ? (d.c1 - d.c0 + 255) / 256 : 0;      // (unsharded: the emit role updates the duty cycle)
    LAUNCH(h, "tm_mid", k_mid_rows, 1 + n_cls + n_sp_rows + n_duty + h->zero_blocks, 256, d, p, n_active, want_winner, learning, n_cls, bank, n_inputs, n_sp_rows, 0, n_duty);
    if (defer_tail) { h->tail_pending = true; h->tail_p = p; return; }
    if (fuse && (front_wmode >= 0 || n_tail_rows)) {
        const int epl = learn_epl(d), n_learn = h->lean_learn_blocks, n_scan = h->lean_scan_blocks, spec = scan_spec_blocks(h);
        const size_t lds = std::max(std::max(learn_lds(epl, 256), scan_lds(d, 1)), (size_t)SEL_BINS * 4);
        const int grid = n_learn + n_scan + (n_tail_rows ? n_tail_rows : h->lean_overlap_blocks);
        const char *name = n_tail_rows ? "tm_learn+tm_scan+sp_learn" : "tm_learn+tm_scan+shard_overlap";
#define LAUNCH_LST(E_) LAUNCH_ON(h, h->stream, lds, name, (k_learn_scan_tail<E_>), grid, 256, d, p, n_learn, n_scan, spec, bank, n_inputs, h->G, front_wmode, n_tail_rows)
        switch (epl) { case 1: LAUNCH_LST(1); break; case 2: LAUNCH_LST(2); break; case 4: LAUNCH_LST(4); break; default: LAUNCH_LST(8); break; }
#undef LAUNCH_LST
        return;
    }
    enqueue_tm_tail(h, p);
    // (a large pool streams through kernels of its own: the front as a launch behind them)
    if (front_wmode >= 0)
        LAUNCH(h, "shard_overlap", k_shard_overlap, h->sp_blocks, RB, d, bank, n_inputs, h->G, p, front_wmode, 1);
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
    return h->cfg.enable_sp && h->cfg.enable_tm && h->world == 1 && h->emit_fused_open && h->d.sel_passes == 2;
}

// the three-launch schedule (htm_pipeline.h): the scan's LDS bitmap, one select histogram, the learning role and the
// scan in one launch.  BITHTM_LEAN=0: the four-launch schedule below.
static bool can_lean(const htm_handle *h) {
    return h->knob_lean && can_pipeline(h) && scan_lds(h->d, 1) <= 64 * 1024 && h->emit_fits_lean;
}

// the last launch of a step in the two- and three-launch schedules: learn(p) + scan(p) beside the select finish + winner list of the
// step of the other parity (n_emit blocks of it, or none)
static void launch_learn_scan_emit(htm_handle *h, int p, int n_emit) {
    Dev &d = h->d;
    const int epl = learn_epl(d);
    const size_t lds = std::max(std::max(learn_lds(epl, 256), lean_scan_lds(d)), sizeof(EmitShared));
    // A large pool streams.  DYN (the default): the grid is what is resident at once and every block ends up scanning
    // (role_scan); the kernel is told by the sign of its n_scan argument.  More learning blocks than a small pool gets: a wave
    // per work item (a large learned pool has ~2 800 a step), so that no block joins late because its waves had second items.
    // BITHTM_SCAN_DYN=0: scan blocks with fixed shares -- more of them than are resident at once (as the select finish's and the
    // learning role's blocks leave, the dispatcher fills their slots; the 768 of the small-pool form left the launch 13 % longer).
    const bool large = scan_pool_is_large(h), dyn = large && h->knob_scan_dyn > 0;
    const int n_learn = dyn ? h->lean_learn_blocks_large : h->lean_learn_blocks;
    int n_scan = large ? h->lean_scan_blocks_large : h->lean_scan_blocks;
    if (dyn) n_scan = std::max(64, h->lean_resident_large - n_emit - n_learn);
    const int grid = n_emit + n_learn + n_scan;
    if (large && !dyn) n_scan = -n_scan;
    const int spec = dyn ? 0 : scan_spec_blocks(h);
    // (the launch's name says which form of the scan it holds: htm_profile_read is how tests and bench.py tell)
    const char *lse_name = scan_pool_is_large(h) ? "tm_learn+tm_scan_large+sp_emit" : "tm_learn+tm_scan+sp_emit";
#define LAUNCH_LSE(EPL_, MINW_, TAB_) LAUNCH_ON(h, h->stream, lds, lse_name, (k_learn_scan_emit<EPL_, MINW_, TAB_>), grid, 256, d, p, n_emit, n_learn, n_scan, spec)
    if (scan_pool_is_large(h)) {
        // (the streaming form reads the cell words from memory: with the LDS tables of the small-pool form it measured 3 % slower --
        // the LDS pipe is one of the things that bound it)
        switch (epl) { case 1: LAUNCH_LSE(1, 4, false); break; case 2: LAUNCH_LSE(2, 4, false); break; case 4: LAUNCH_LSE(4, 4, false); break; default: LAUNCH_LSE(8, 4, false); break; }
    } else if (lean_tab(d)) {
        switch (epl) { case 1: LAUNCH_LSE(1, 6, true); break; case 2: LAUNCH_LSE(2, 6, true); break; case 4: LAUNCH_LSE(4, 6, true); break; default: LAUNCH_LSE(8, 6, true); break; }
    } else {
        switch (epl) { case 1: LAUNCH_LSE(1, 6, false); break; case 2: LAUNCH_LSE(2, 6, false); break; case 4: LAUNCH_LSE(4, 6, false); break; default: LAUNCH_LSE(8, 6, false); break; }
    }
#undef LAUNCH_LSE
}

// sp_done: the winner list of this step exists (the previous step's last launch, or the cold start).  next_sp: select
// the next step's winners beside this step's Temporal Memory.
static void enqueue_lean(htm_handle *h, int p, int learning, const uint32_t *bank, int n_inputs, StepPlan plan) {
    Dev &d = h->d;
    const int n_act = (d.k * d.KP + 255) / 256, n_rows = learning ? d.k : 0;
    const int n_cls = learning ? kClassifyBlocks : 0, n_ov = plan.next_sp ? h->lean_overlap_blocks : 0;
    if (h->knob_lean == 2) {                        // the two-launch schedule: both of these in one (htm_pipeline.h)
        // (a large pool has a classification block read ~250 match words and classify tens of them: more blocks there, though they
        // are not all resident from the start -- 350-pattern pool of the bench, 1.6 M segments: 32 blocks 11.7 k timesteps/s, 128: 13.3, 256: 13.3,
        // 384: 13.1; three launches: 12.8)
        const int n_cls2 = !learning ? 0 : (scan_pool_is_large(h) && !h->lean2_classify_set) ? std::max(h->lean2_classify_blocks, 192) : h->lean2_classify_blocks;
        const int n_duty = n_ov ? 0 : h->c256_blocks, n_clear = d.WPC * h->c256_blocks;
        LAUNCH_ON(h, h->stream, (size_t)SEL_BINS * 4, "tm_activate+tm_mid+sp_learn+sp_overlap", k_act_mid_rows,
                  n_act + 1 + n_cls2 + n_rows + n_ov + n_duty + n_clear + h->zero_blocks, 256, d, p, d.k, n_act, learning, n_cls2, bank, n_inputs, n_rows, h->G, n_ov,
                  n_duty, n_clear, h->lean2_order);
    } else {
        LAUNCH(h, "tm_activate+sp_learn", k_act_rows, n_act + n_rows + (1 + d.WPC) * h->c256_blocks, 256, d, p, d.k, n_act, bank, n_inputs, n_rows, h->c256_blocks);
        LAUNCH_ON(h, h->stream, (size_t)SEL_BINS * 4, "tm_mid+sp_overlap", k_mid_overlap, 1 + n_cls + n_ov + h->zero_blocks, 256, d, p, d.k, 1, learning, n_cls,
                  bank, n_inputs, h->G, n_ov);
    }
    launch_learn_scan_emit(h, p, plan.next_sp ? h->c256_blocks : 0);
}

// the four launches of a pipelined step (see the kernels): step p's Temporal Memory beside SP work of
// the following steps
static void enqueue_pipelined(htm_handle *h, int p, int learning, const uint32_t *bank, int n_inputs, StepPlan plan) {
    Dev &d = h->d;
    const int n_cls = learning ? kClassifyBlocks : 0;
    const int n_emit = plan.next_sp ? h->c256_blocks : 0;
    LAUNCH_ON(h, h->stream, sizeof(EmitShared), "tm_activate+sp_emit", k_open_emit, n_emit + (d.k * d.KP + 255) / 256, 256, d, p, n_emit, d.k);
    const int n_rows = (plan.next_sp && learning) ? d.k : 0, n_duty = plan.next_sp ? h->c256_blocks : 0;
    LAUNCH(h, "tm_mid+sp_learn", k_mid_rows, 1 + n_cls + n_rows + n_duty + h->zero_blocks, 256, d, p, d.k, 1, learning, n_cls, bank, n_inputs, n_rows, 1, n_duty);
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
        if (use_lds) LAUNCH_ON(h, h->stream, lds, "tm_scan_large+sp_select", (k_scan_sel<true, 1>), grid, 256, d, p, n_sel, n_clear, p, spec);
        else LAUNCH_ON(h, h->stream, lds, "tm_scan_large+sp_select", (k_scan_sel<false, 1>), grid, 256, d, p, n_sel, n_clear, p, spec);
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
    const int wmode = step_wmode(h);
    if (can_lean(h)) {                              // the winner list of this step, nothing else
        enqueue_sp_front(h, bank, n_inputs, p, wmode);
        enqueue_sp_back(h, bank, n_inputs, p, 1, 0, false, wmode);
        return;
    }
    enqueue_sp_front(h, bank, n_inputs, p, wmode);
    enqueue_sp_back(h, bank, n_inputs, p, 1, EMIT_DUTY | EMIT_CLEAR, learning != 0, wmode);
    LAUNCH(h, "sp_overlap", k_sp_overlap, h->sp_blocks, RB, d, bank, n_inputs, h->G, p, p ^ 1, 1, 0);
    LAUNCH(h, "sp_select", k_sel_pass, h->sel_blocks, RB, d, 1, p ^ 1);
}

static void enqueue_rest(htm_handle *h, int p, const uint32_t *bank, int n_inputs, int learning, StepPlan plan) {
    if ((plan.sp_done || plan.next_sp) && can_lean(h)) {
        enqueue_lean(h, p, learning, bank, n_inputs, plan);
    } else if (plan.sp_done || plan.next_sp) {
        enqueue_pipelined(h, p, learning, bank, n_inputs, plan);
    } else {                                        // one role per launch
        enqueue_sp_back(h, bank, n_inputs, p, 1, EMIT_ALL, false, step_wmode(h));
        enqueue_tm(h, h->d.k, learning, 1, p, bank, n_inputs, true);
    }
}

static int enqueue_step(htm_handle *h, const uint32_t *bank, int n_inputs, int learning, StepPlan plan, const uint32_t *host_input = nullptr) {
    if (!plan.sp_done && !plan.next_sp) enqueue_sp_front(h, bank, n_inputs, (int)(h->step_host & 1), step_wmode(h), host_input);
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
    {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        g_registry.erase(std::remove(g_registry.begin(), g_registry.end(), h), g_registry.end());
    }
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    for (auto &kv : h->graphs) hipGraphExecDestroy(kv.second);
    for (auto &kv : h->shard_graphs) hipGraphExecDestroy(kv.second);
    for (auto &v : h->prof_events)
        for (auto &pr : v) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (hipEvent_t e : h->prof_all) hipEventDestroy(e);
    if (h->rccl_comm && g_rccl_destroy) g_rccl_destroy(h->rccl_comm);
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
        if (cfg->cell_dim < 1 || cfg->cell_dim > 64) return fail_create(nullptr, "htm_create: cell_dim must be in 1..64", HTM_ERR_ARGUMENT);
        if (cfg->segment_slots < 64 || cfg->segment_slots > MAX_SLOTS || cfg->segment_slots % 64)
            return fail_create(nullptr, "htm_create: segment_slots must be a multiple of 64 in 64..512", HTM_ERR_ARGUMENT);
        if (cfg->segment_capacity < 1) return fail_create(nullptr, "htm_create: segment_capacity < 1", HTM_ERR_ARGUMENT);
        if (cfg->segment_sampling_synapses < 1 || cfg->segment_sampling_synapses > 64)
            return fail_create(nullptr, "htm_create: segment_sampling_synapses must be in 1..64", HTM_ERR_ARGUMENT);
        if (cfg->segment_activation_threshold < cfg->segment_matching_threshold)      // projections.py:211
            return fail_create(nullptr, "htm_create: activation threshold < matching threshold", HTM_ERR_ARGUMENT);
        if ((long long)cfg->column_dim * (cfg->cell_dim > 32 ? 64 : 32) > 0x7FFFFFFFLL) return fail_create(nullptr, "htm_create: column_dim too large", HTM_ERR_ARGUMENT);
    }
    if (!cfg->enable_sp && !cfg->enable_tm) return fail_create(nullptr, "htm_create: nothing enabled", HTM_ERR_ARGUMENT);
    const int world = cfg->shard_world > 1 ? cfg->shard_world : 1;
    if (world > 1) {
        if (!cfg->enable_sp || !cfg->enable_tm) return fail_create(nullptr, "htm_create: a sharded handle needs SP and TM", HTM_ERR_ARGUMENT);
        if (cfg->cell_dim > 32) return fail_create(nullptr, "htm_create: a sharded handle takes cell_dim up to 32 (the exchange record carries one 32-bit word per column)", HTM_ERR_ARGUMENT);
        if (world > 64) return fail_create(nullptr, "htm_create: at most 64 shards", HTM_ERR_ARGUMENT);
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
    h->window_known = false;
    h->shard_front_wmode = 0;
    h->shard_graph_ok = false;
    h->rccl_comm = nullptr;
    h->shard_send = h->shard_recv = nullptr;
    h->phase_active = 0;
    h->phase_open = false;
    h->import_keep = false;
    h->ahead_bank = nullptr;
    h->ahead_n_inputs = h->ahead_learning = 0;
    h->ahead_lean = false;
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
    d.KP = d.K > 32 ? 64 : 32;
    d.LK = d.K > 32 ? 6 : 5;
    d.WPC = d.KP / 32;
    d.k = cfg->active_columns;
    d.E = cfg->enable_tm ? cfg->segment_slots : 64;
    d.Scap = cfg->enable_tm ? cfg->segment_capacity : 0;
    d.Lcap = d.Scap;
    if (world > 1) {
        const long long dflt = std::min<long long>(d.Scap, 2LL * d.Scap / world + 1024);
        d.Lcap = cfg->segment_capacity_local > 0 ? std::min(cfg->segment_capacity_local, cfg->segment_capacity) : (int)dflt;
    }
    d.work_cap = d.Lcap + d.k * d.KP;
    // the select: all columns and their k largest; a shard selects its own candidates for the exchange
    d.sel_lo = d.c0; d.sel_hi = d.c1;
    d.sel_k = d.n_cand = std::min(d.k, d.c1 - d.c0);
    d.cand_cap = shard_cand_cap(d.n_cand, d.c1 - d.c0);
    d.hot_budget = std::max(1, std::min(d.cand_cap, (SHARD_HOT_KEYS * 1024) / std::max(world, 1)));
    d.hot_target = std::max(1, std::min(d.sel_k, d.hot_budget / 2));
    d.sp_thr = cfg->sp_permanence_threshold; d.sp_don = cfg->sp_delta_on; d.sp_doff = cfg->sp_delta_off;
    d.coef = cfg->boost_coefficient; d.mom = cfg->duty_momentum; d.dinc = cfg->duty_increment;
    d.lrn_act = cfg->tm_learn_active; d.lrn_inact = cfg->tm_learn_inactive;
    d.pun_act = cfg->tm_punish_active; d.pun_inact = cfg->tm_punish_inactive;
    d.lrn_prune = cfg->tm_learn_prune; d.pun_prune = cfg->tm_punish_prune;
    d.perm_init = cfg->tm_permanence_initial; d.perm_thr = cfg->tm_permanence_threshold;
    d.eps = EPS32;
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
        rc |= dalloc(h, &d.hist0, (size_t)2 * HIST0_PAR);
        rc |= dalloc(h, &d.sel_blk, (C + 255) / 256);
        rc |= dalloc(h, &d.sel_rec, (C + 255) / 256 * 32);
        rc |= dalloc(h, &d.input_stage, (size_t)d.W);
    }
    if (cfg->enable_tm) {
        const size_t S = d.Lcap, E = d.E, G = d.Scap;        // local rows (= ids on an unsharded handle), slots, segment ids
        const size_t KP = d.KP, WPC = d.WPC;
        for (int q = 0; q < 2; ++q) {
            rc |= dalloc(h, &d.act[q], C * WPC);
            rc |= dalloc(h, &d.pred[q], C * WPC);
            rc |= dalloc(h, &d.winners[q], k * KP);
        }
        rc |= dalloc(h, &d.win[0], C * WPC);
        rc |= dalloc(h, &d.win[1], C * WPC);
        d.colwords = (int)((C + 63) / 64) * 2;
        // (the emit blocks write the bitmap words of whole 256-column blocks)
        const size_t colwords_padded = (size_t)((C + 255) / 256) * 8;
        rc |= dalloc(h, &d.colbits[0], colwords_padded);
        rc |= dalloc(h, &d.colbits[1], colwords_padded);
        rc |= dalloc(h, &d.bursting, k);
        rc |= dalloc(h, &d.actw_id, k * WPC + 8);
        rc |= dalloc(h, &d.winw_idx, k * WPC + 8);
        rc |= dalloc(h, &d.actcnt, k * WPC + 8);
        rc |= dalloc(h, &d.act_list, k * WPC + 16);   // (staged into LDS in 16-byte units)
        rc |= dalloc(h, &d.col_rank[0], colwords_padded);
        rc |= dalloc(h, &d.col_rank[1], colwords_padded);
        rc |= dalloc(h, &d.unacc_word, k * WPC + 8);
        rc |= dalloc(h, &d.fan, (size_t)2 * FAN_COUNTERS * FAN_STRIDE);
        rc |= dalloc(h, &d.unacc_list, k * KP);
        rc |= dalloc(h, &d.seg_cell, S);
        rc |= dalloc(h, &d.seg_nsyn, S);
        rc |= dalloc(h, &d.presyn, S * E);
        rc |= dalloc(h, &d.sperm, S * E);
        rc |= dalloc(h, &d.segcount, C * KP);
        for (int q = 0; q < 2; ++q) {
            rc |= dalloc(h, &d.cellmax[q], C * KP);
            rc |= dalloc(h, &d.match_bits[q], (S + 255) / 256 * 8);
        }
        rc |= dalloc(h, &d.seg_info, S);
        rc |= dalloc(h, &d.seg_jit, S);
        rc |= dalloc(h, &d.work, (size_t)d.work_cap);
        rc |= dalloc(h, &d.recyc_cnt, (G + 1023) / 1024);
        rc |= dalloc(h, &d.recyc_cnt2, ((G + 1023) / 1024 + 1023) / 1024 + 1);
        rc |= dalloc(h, &d.recyc_need, 2 * k * KP);
        rc |= dalloc(h, &d.dead_list, (size_t)1 + DEAD_CAP);
        if (world > 1) {
            rc |= dalloc(h, &d.seg_gid, S);
            rc |= dalloc(h, &d.g2l, G);
            rc |= dalloc(h, &d.dead_bits, (G + 31) / 32 + 32);
            rc |= dalloc(h, &d.lfree, S);
            rc |= dalloc(h, &d.asg_gid, k * 32);
            rc |= dalloc(h, &d.cand_cols, (size_t)d.n_cand + 8);
            if (!rc && (hipMemsetAsync(d.seg_gid, 0xFF, S * 4, h->stream) != hipSuccess ||
                        hipMemsetAsync(d.g2l, 0xFF, G * 4, h->stream) != hipSuccess)) { h->err = "hipMemsetAsync failed"; rc = HTM_ERR_HIP; }
        }
        rc |= dalloc(h, &h->d_cols_stage, k);
    }
    if (rc) return fail_create(h, h->err, HTM_ERR_HIP);
    // lanes per SP row: the smallest power of two >= W4, at most 64
    h->G = 1;
    h->seg_hint = 0;
    h->graph_steps = 16;
    if (const char *e = getenv("BITHTM_GRAPH_STEPS")) h->graph_steps = std::max(1, std::min(256, atoi(e)));
    h->eager_below = 64;
    if (const char *e = getenv("BITHTM_EAGER_BELOW")) h->eager_below = std::max(0, atoi(e));
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
    h->sel_blocks = std::max(1, std::min((d.sel_hi - d.sel_lo + RB - 1) / RB, 128));
    h->c256_blocks = (d.sel_hi - d.sel_lo + 255) / 256;        // blocks of the emit role: 256 columns of the select's range each
    h->s1024_blocks = std::max(1, (d.Scap + 1023) / 1024);
    h->scan_blocks = std::max(1, std::min((d.Lcap + SCAN_SEGS - 1) / SCAN_SEGS, 2048));
    if (h->scan_blocks > 256) h->scan_blocks = (h->scan_blocks + 255) & ~255;     // (role_scan: whole groups of 256 blocks)
    h->zero_blocks = std::max(1, std::min((d.Lcap / 128 + 4095) / 4096, 1024));     // k_mid_rows: 16 stores of 16 bytes per thread at most
    // the three-launch schedule: waves of one 256-thread block per work item in the steady state; the scan's waves take
    // two groups of segments each, so that emit + learn + scan are all resident at once (tuning knobs)
    // launches per step of htm_run's pipelined schedule -- 2 (the default): two (k_act_mid_rows + k_learn_scan_emit); 1: three; 0: four
    h->knob_lean = getenv("BITHTM_LEAN") ? std::max(0, std::min(2, atoi(getenv("BITHTM_LEAN")))) : 2;
    h->lean2_classify_blocks = getenv("BITHTM_LEAN2_CLASSIFY") ? std::max(1, atoi(getenv("BITHTM_LEAN2_CLASSIFY"))) : kClassifyBlocks;
    h->lean2_classify_set = getenv("BITHTM_LEAN2_CLASSIFY") != nullptr;
    // the host-fed step (htm_step): select finish alone + k_act_mid_rows, instead of select finish with the activation in its blocks +
    // k_mid_rows (BITHTM_STEP_SPLIT=0: as before)
    h->knob_step_split = getenv("BITHTM_STEP_SPLIT") ? atoi(getenv("BITHTM_STEP_SPLIT")) != 0 : 1;
    {   // the grid order of k_act_mid_rows' roles (hex digits: 0 activation, 1 middle, 2 rows, 3 overlap); a permutation with 0 before 1
        // (measured at the bench shape, 8 waves per SIMD: 0312 42.4 k timesteps/s, 0321 41.9, 0123 41.2, 3201 40.2)
        const int o = getenv("BITHTM_LEAN2_ORDER") ? (int)strtol(getenv("BITHTM_LEAN2_ORDER"), nullptr, 16) : 0x0312;
        int seen = 0, pos[4] = {-1, -1, -1, -1};
        for (int i = 0; i < 4; ++i) { const int r = (o >> (4 * (3 - i))) & 15; if (r < 4) { seen |= 1 << r; pos[r] = i; } }
        h->lean2_order = (o >= 0 && o <= 0x3333 && seen == 15 && pos[0] < pos[1]) ? o : 0x0312;
    }
    h->knob_fuse_tm = getenv("BITHTM_FUSE_TM") ? atoi(getenv("BITHTM_FUSE_TM")) != 0 : 1;
    h->knob_shard_window = getenv("BITHTM_SHARD_WINDOW") ? atoi(getenv("BITHTM_SHARD_WINDOW")) != 0 : 1;
    h->knob_scan_large = getenv("BITHTM_SCAN_LARGE") ? atoi(getenv("BITHTM_SCAN_LARGE")) : -1;
    // (test knob: a small model crosses the threshold in the middle of a run, as the headline shape does with more patterns)
    h->scan_large_above = getenv("BITHTM_SCAN_LARGE_ABOVE") ? std::max(0, atoi(getenv("BITHTM_SCAN_LARGE_ABOVE"))) : 3 * 1536 * SCAN_SEGS;
    h->knob_step_window = getenv("BITHTM_STEP_WINDOW") ? atoi(getenv("BITHTM_STEP_WINDOW")) != 0 : 1;
    h->knob_tail_rows = getenv("BITHTM_TAIL_ROWS") ? atoi(getenv("BITHTM_TAIL_ROWS")) != 0 : 1;
    // (0: scan blocks with fixed shares, nothing joining; else every block of the launch joins the scan)
    h->knob_scan_dyn = getenv("BITHTM_SCAN_DYN") ? atoi(getenv("BITHTM_SCAN_DYN")) != 0 : 1;
    h->knob_defer_tail = getenv("BITHTM_DEFER_TAIL") ? atoi(getenv("BITHTM_DEFER_TAIL")) != 0 : 1;
    h->tail_pending = false;
    h->tail_p = 0;
    h->lean_overlap_blocks = getenv("BITHTM_LEAN_OVERLAP") ? std::max(1, atoi(getenv("BITHTM_LEAN_OVERLAP"))) : h->sp_blocks * (RB / 256);
    h->lean_learn_blocks = getenv("BITHTM_LEAN_LEARN") ? std::max(1, atoi(getenv("BITHTM_LEAN_LEARN"))) : 512;
    h->lean_learn_blocks_large = getenv("BITHTM_LEAN_LEARN_LARGE") ? std::max(1, atoi(getenv("BITHTM_LEAN_LEARN_LARGE"))) : getenv("BITHTM_LEAN_LEARN") ? h->lean_learn_blocks : 768;
    h->lean_scan_blocks = getenv("BITHTM_LEAN_SCAN") ? std::max(1, atoi(getenv("BITHTM_LEAN_SCAN"))) : (h->scan_blocks > 512 ? std::max(256, (h->scan_blocks * 3 / 8 + 255) & ~255) : h->scan_blocks);
    h->lean_scan_blocks_large = getenv("BITHTM_LEAN_SCAN_LARGE") ? std::max(1, atoi(getenv("BITHTM_LEAN_SCAN_LARGE"))) : getenv("BITHTM_LEAN_SCAN") ? h->lean_scan_blocks : h->scan_blocks;
    {
        hipDeviceProp_t prop;
        h->cus = hipGetDeviceProperties(&prop, h->device) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    if (const char *e = getenv("BITHTM_SCAN_BLOCKS")) h->scan_blocks = std::max(1, atoi(e));      // tuning knob
    // boosted = float32 factor x integer overlap <= input_dim has at most 24 + bit_length(I)
    // significant bits, so the low 53 - 24 - bit_length(I) bits of every key are zero and the
    // radix passes that would only see them are skipped.
    {
        int B = 0;
        while ((1ll << B) <= (long long)d.I) ++B;
        const int informative = std::min(64, 64 - (29 - B) - KEY_SHIFT);      // (select_key moves the bits up by KEY_SHIFT)
        d.sel_passes = std::max(1, std::min(SEL_MAX_PASSES, (informative + SEL_DIGIT - 1) / SEL_DIGIT));
        d.low_zero = 64 - informative;            // key bits [0, low_zero) are zero in every key
        // Emit grids whose blocks are all resident at once finish the select inside k_sp_emit (two digits
        // by launches, the rest through the record exchange, in which blocks wait for each other).  What
        // fits is asked of the runtime, kernel by kernel, not assumed.
        {
            const int c256 = (d.sel_hi - d.sel_lo + 255) / 256;
            hipDeviceProp_t prop;
            int per_cu_emit = 0, per_cu_open = 0;
            if (hipGetDeviceProperties(&prop, h->device) == hipSuccess &&
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_emit, (const void *)k_sp_emit, 256, 0) == hipSuccess &&
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_open, (const void *)k_open_emit, 256, sizeof(EmitShared)) == hipSuccess) {
                const int cus = prop.multiProcessorCount;
                h->emit_fits = c256 <= std::min(1024, per_cu_emit * cus);
                // the pipelined launch puts the activation blocks of the current step behind the emit blocks
                h->emit_fits_open = h->emit_fits && c256 + (d.k * d.KP + 255) / 256 <= std::min(1024, per_cu_open * cus);
                // the three-launch schedule: the emit blocks come first in the grid of the learn + scan + emit kernel
                // (asked of every instantiation enqueue_lean may launch for this handle -- LDS tables or not, small-pool or
                // large-pool scan: they differ in launch bounds and registers -- and the smallest answer counts)
                int per_cu_lean = 1 << 30;
                h->lean_resident_large = 1 << 30;
                const size_t lean_lds = std::max(std::max(learn_lds(learn_epl(d), 256), lean_scan_lds(d)), sizeof(EmitShared));
                const int epl = learn_epl(d);
#define LSE_VARIANTS(E_) {(const void *)k_learn_scan_emit<E_, 6, true>, (const void *)k_learn_scan_emit<E_, 6, false>, (const void *)k_learn_scan_emit<E_, 4, false>}
                const void *kerns[4][3] = {LSE_VARIANTS(1), LSE_VARIANTS(2), LSE_VARIANTS(4), LSE_VARIANTS(8)};
#undef LSE_VARIANTS
                bool asked = cfg->enable_tm && h->emit_fits && lean_lds <= 64 * 1024;
                for (int v = 0; asked && v < 3; ++v) {
                    if (v == 0 && !lean_tab(d)) continue;           // (never launched without the tables)
                    int per_cu = 0;
                    asked = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kerns[epl == 1 ? 0 : epl == 2 ? 1 : epl == 4 ? 2 : 3][v], 256, lean_lds) == hipSuccess;
                    per_cu_lean = std::min(per_cu_lean, per_cu);
                    if (asked && v == 2) h->lean_resident_large = per_cu * cus;      // (the large-pool form)
                }
                h->emit_fits_lean = asked && c256 <= std::min(1024, per_cu_lean * cus);
                if (h->lean_resident_large == (1 << 30)) h->lean_resident_large = 4 * cus;
            } else {
                (void)hipGetLastError();
                h->emit_fits = h->emit_fits_open = h->emit_fits_lean = false;
                h->lean_resident_large = 1024;
            }
        }
        if (!getenv("BITHTM_LEAN2_CLASSIFY")) {
            // the two-launch schedule's first launch: as many classification blocks as are resident BESIDE the activation, the
            // overlap and the winner rows -- a block that waits for a slot starts a round late, and the middle role's blocks wait for
            // the activation whenever they start (bench shape: 164 + 512 + 1 311 + 1 of 2 048 slots leave 60; small models get 384)
            hipDeviceProp_t prop;
            int per_cu = 0, resident = 1536;
            if (hipGetDeviceProperties(&prop, h->device) == hipSuccess &&
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_act_mid_rows, 256, (size_t)SEL_BINS * 4) == hipSuccess)
                resident = per_cu * prop.multiProcessorCount;
            else
                (void)hipGetLastError();
            const int others = (d.k * d.KP + 255) / 256 + h->lean_overlap_blocks + d.k + 1;
            // (... but no more than 32 where they leave fewer than 128: measured at the bench shape, 60 free slots -- 32 blocks 42.4 k
            // timesteps/s, 40: 41.9, 56: 41.6)
            const int room = (resident - others) & ~7;
            h->lean2_classify_blocks = room >= 128 ? std::min(kClassifyBlocks, room) : 32;
        }
        h->sel_passes_full = d.sel_passes;
        h->sel_passes_fused = std::min(d.sel_passes, 2);
        // test knobs: more launched digits (smaller buckets); fewer record slots (forces the fallback)
        if (const char *e = getenv("BITHTM_SEL_LAUNCH_DIGITS")) h->sel_passes_fused = std::max(2, std::min(d.sel_passes, atoi(e)));
        d.cand_d = CAND_D;
        if (const char *e = getenv("BITHTM_CAND_D")) d.cand_d = std::max(0, std::min(CAND_D, atoi(e)));
        d.cand_pairwise = CAND_PAIRWISE;
        if (const char *e = getenv("BITHTM_CAND_PAIRWISE")) d.cand_pairwise = std::max(0, atoi(e));     // test knobs
        d.cls_rows_max = getenv("BITHTM_CLASSIFY_WORDS_ABOVE") ? std::max(0, atoi(getenv("BITHTM_CLASSIFY_WORDS_ABOVE"))) : -1;
        d.win_offset = getenv("BITHTM_SEL_WINDOW_OFFSET") ? std::max(0, atoi(getenv("BITHTM_SEL_WINDOW_OFFSET"))) : 0;
        d.cand_zoom = getenv("BITHTM_CAND_ZOOM") ? atoi(getenv("BITHTM_CAND_ZOOM")) : -1;      // (test knob: the pairs above which a merge is cut to a sub-bin; -1 = more pairs than blocks by a quarter)
        d.cand_speculate = getenv("BITHTM_CAND_SPECULATE") ? atoi(getenv("BITHTM_CAND_SPECULATE")) != 0 : 1;     // (test knob: 0 = always the general path)
        d.cand_take_all = getenv("BITHTM_CAND_TAKE_ALL") ? atoi(getenv("BITHTM_CAND_TAKE_ALL")) != 0 : 1;      // (test knob: 0 = a shard's local select always cuts exactly)
        d.poll_delay = getenv("BITHTM_POLL_DELAY") ? std::max(0, std::min(64, atoi(getenv("BITHTM_POLL_DELAY")))) : 7;     // (the select finish's first look at the other blocks' records: x 256 clocks after its own)
        d.cand_others = CAND_OTHERS;
        if (const char *e = getenv("BITHTM_CAND_OTHERS")) d.cand_others = std::max(0, std::min(CAND_OTHERS, atoi(e)));
    }
    e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail_create(h, std::string("hipStreamSynchronize: ") + hipGetErrorString(e), HTM_ERR_HIP);
    {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        g_registry.push_back(h);
    }
    refresh_exchange_mode(h);
    *out = h;
    return HTM_OK;
}

// TemporalMemory.process(..., epsilon=) (networks.py:91): the tolerance of the "best matching" / "least used" ties
// (networks.py:81,88; projections.py:267), compared as float32.  0 < epsilon <= 1: the other uses (prediction > epsilon,
// potential < epsilon) then mean what they mean at 1e-8.  Kernels get it with their arguments: cached graphs are dropped.
extern "C" int htm_set_epsilon(htm_handle *h, float epsilon) {
    if (!h) return HTM_ERR_ARGUMENT;
    if (h) flush_tail(h);
    if (!(epsilon > 0.f) || epsilon > 1.f) { h->err = "htm_set_epsilon: need 0 < epsilon <= 1"; return HTM_ERR_ARGUMENT; }
    REJECT_WHEN_AHEAD(h);
    if (epsilon == h->d.eps) return HTM_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (auto &kv : h->graphs) hipGraphExecDestroy(kv.second);
    h->graphs.clear();
    h->d.eps = epsilon;
    return HTM_OK;
}

extern "C" int htm_get_stream(htm_handle *h, void **stream) {
    if (!h || !stream) return HTM_ERR_ARGUMENT;
    *stream = (void *)h->stream;
    return HTM_OK;
}

extern "C" int htm_sync(htm_handle *h) {
    if (!h) return HTM_ERR_ARGUMENT;
    if (h) flush_tail(h);
    HIPCHK(h, hipSetDevice(h->device));
    // a short run ends within a millisecond: poll for that long (a blocking wait is woken tens of microseconds late),
    // then block
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(h->stream);
        if (e == hipSuccess) return HTM_OK;
        if (e != hipErrorNotReady) { h->err = std::string("hipStreamQuery: ") + hipGetErrorString(e); return HTM_ERR_HIP; }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
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
    if (h) flush_tail(h);
    int rc = check_rows(h, rows, row_begin, row_count);
    if (rc) return rc;
    REJECT_WHEN_AHEAD(h);
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
    if (h) flush_tail(h);
    int rc = check_rows(h, rows, row_begin, row_count);
    if (rc) return rc;
    REJECT_WHEN_AHEAD(h);
    if (row_count == 0) return HTM_OK;
    Dev &d = h->d;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy2DAsync(rows, (size_t)d.I * 8, d.perm + (size_t)row_begin * d.Ipad, (size_t)d.Ipad * 8,
                               (size_t)d.I * 8, (size_t)row_count, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return HTM_OK;
}

// htm_sp_phase leaves the current (not yet closed) timestep half done: keys, boosted overlaps and the top-digit
// histogram of parity step_host & 1 (DenseProjection.process / ExponentialBoosting.process called on their own on the
// objects of a live SpatialPooler).  An entry point that runs the WHOLE step starts over: the histogram its overlap
// accumulates into must be clean (only a select clears it), and the Spatial Pooler fields htm_read returns are the last
// completed step's again.
static int close_open_phases(htm_handle *h) {
    if (!h->phase_open) return 0;
    const int p = (int)(h->step_host & 1);
    if (h->cfg.enable_sp) HIPCHK(h, hipMemsetAsync(h->d.hist0 + (size_t)p * HIST0_PAR, 0, (size_t)HIST0_PAR * 4, h->stream));
    h->phase_open = false;
    h->phase_active = 0;
    return 0;
}

static int stage_input(htm_handle *h, const uint32_t *packed_input) {
    Dev &d = h->d;
    const int words = (d.I + 31) / 32;
    HIPCHK(h, hipMemcpyAsync(d.input_stage, packed_input, (size_t)words * 4, hipMemcpyHostToDevice, h->stream));
    return 0;
}

extern "C" int htm_step(htm_handle *h, const uint32_t *packed_input, int32_t learning) {
    if (!h || !packed_input) return HTM_ERR_ARGUMENT;
    REJECT_WHEN_AHEAD(h);
    if (!h->cfg.enable_sp || !h->cfg.enable_tm) { h->err = "htm_step needs a handle with SP and TM"; return HTM_ERR_STATE; }
    if (h->world > 1) { h->err = "sharded handle: use htm_shard_begin / htm_shard_finish"; return HTM_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    refresh_exchange_mode(h);
    int rc = close_open_phases(h);
    if (rc) return rc;
    if (h->d.W > ARG_INPUT_WORDS) {                 // (an input too wide for the launch's arguments: staged by a copy)
        flush_tail(h);
        rc = stage_input(h, packed_input);
        if (rc) return rc;
    }
    // Three launches per step for a caller that steps and steps: the learning role and the scan of a step need nothing of the
    // NEXT input and nothing the next step's overlap touches -- they are held back and ride beside that overlap (the
    // permanence rows, which the overlap does read, in the middle launch instead).  Any other call lets them go first.
    Dev &d = h->d;
    if (h->knob_defer_tail && d.W <= ARG_INPUT_WORDS && tm_tail_fused(h) && !h->profile) {
        const int p = (int)(h->step_host & 1), wmode = step_wmode(h);
        // the held-back learning role and scan of the step before ride beside THIS step's select finish (k_learn_scan_emit, the last
        // launch of htm_run's schedules) where that launch is available: the overlap then has the first launch to itself (4.5 us), and the
        // select finish -- 7 us of a chain on 256 blocks -- no longer has the GPU to itself.  (The activation, which reads the
        // predictions that scan leaves, has moved behind it: k_act_mid_rows.)
        const bool ride_emit = h->knob_step_split && h->tail_pending && wmode && h->emit_fused && can_lean(h);
        bool emitted = false;
        if (ride_emit) {
            enqueue_sp_front(h, d.input_stage, 1, p, wmode, packed_input);
            h->tail_pending = false;
            launch_learn_scan_emit(h, h->tail_p, h->c256_blocks);
            h->window_known = true;
            emitted = true;
        } else if (h->tail_pending) {
            PackedInputArg in;
            memset(&in, 0, sizeof(in));
            memcpy(in.w, packed_input, (size_t)((d.I + 31) / 32) * 4);
            const int epl = learn_epl(d), n_learn = h->lean_learn_blocks, n_scan = h->lean_scan_blocks, spec = scan_spec_blocks(h);
            const size_t lds = std::max(std::max(learn_lds(epl, 256), scan_lds(d, 1)), (size_t)(SEL_BINS + ARG_INPUT_WORDS) * 4);
            const int grid = n_learn + n_scan + h->lean_overlap_blocks;
            h->tail_pending = false;
#define LAUNCH_LSF(E_) LAUNCH_ON(h, h->stream, lds, "tm_learn+tm_scan+sp_overlap", (k_learn_scan_front<E_>), grid, 256, d, h->tail_p, n_learn, n_scan, spec, in, h->G, p, wmode)
            switch (epl) { case 1: LAUNCH_LSF(1); break; case 2: LAUNCH_LSF(2); break; case 4: LAUNCH_LSF(4); break; default: LAUNCH_LSF(8); break; }
#undef LAUNCH_LSF
            if (!wmode)
                for (int pass = 1; pass < d.sel_passes; ++pass) LAUNCH(h, "sp_select", k_sel_pass, h->sel_blocks, RB, d, pass, p);
        } else {
            enqueue_sp_front(h, d.input_stage, 1, p, wmode, packed_input);
        }
        if (h->knob_step_split) {
            // the select finish on its own (winner list and column bitmap, nothing else), then the two-launch schedule's first launch
            // in the form its last step takes: activation -> fan-in -> middle role beside the winner rows, the duty cycle and the
            // clears (k_act_mid_rows without an overlap role).  The activation no longer waits at the end of the select finish's
            // blocks with nothing beside it, and the rows stream under the Temporal Memory's chain (measured: DESIGN.md section 4)
            if (!emitted) enqueue_sp_back(h, d.input_stage, 1, p, 1, 0, false, wmode);
            const int lrn = learning ? 1 : 0;
            const int n_act = (d.k * d.KP + 255) / 256, n_rows = lrn ? d.k : 0;
            const int n_cls2 = !lrn ? 0 : (scan_pool_is_large(h) && !h->lean2_classify_set) ? std::max(h->lean2_classify_blocks, 192) : h->lean2_classify_blocks;
            const int n_duty = h->c256_blocks, n_clear = d.WPC * h->c256_blocks;
            LAUNCH_ON(h, h->stream, 0, "tm_activate+tm_mid+sp_learn", k_act_mid_rows, n_act + 1 + n_cls2 + n_rows + n_duty + n_clear + h->zero_blocks, 256,
                      d, p, d.k, n_act, lrn, n_cls2, d.input_stage, 1, n_rows, h->G, 0, n_duty, n_clear, h->lean2_order);
            h->tail_pending = true;
            h->tail_p = p;
        } else {
            enqueue_sp_back(h, d.input_stage, 1, p, 1, EMIT_ALL, false, wmode);
            enqueue_tm(h, d.k, learning ? 1 : 0, 1, p, d.input_stage, 1, true, -1, true);
        }
        h->step_host += 1;
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { h->err = std::string("kernel launch: ") + hipGetErrorString(e); return HTM_ERR_HIP; }
        return HTM_OK;
    }
    flush_tail(h);
    return enqueue_step(h, h->d.input_stage, 1, learning ? 1 : 0, StepPlan{false, false, false}, packed_input);
}

extern "C" int htm_sp_step(htm_handle *h, const uint32_t *packed_input, int32_t learning) {
    if (!h || !packed_input) return HTM_ERR_ARGUMENT;
    if (h) flush_tail(h);
    REJECT_WHEN_AHEAD(h);
    if (!h->cfg.enable_sp) { h->err = "handle has no Spatial Pooler"; return HTM_ERR_STATE; }
    // a handle that also owns a Temporal Memory steps both layers together: an SP-only step would skip the SP
    // learning that rides in the TM's middle launch and leave the TM's parity buffers one step behind
    if (h->cfg.enable_tm) { h->err = "htm_sp_step: the handle also has a Temporal Memory; use htm_step"; return HTM_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    refresh_exchange_mode(h);
    int rc = close_open_phases(h);
    if (rc) return rc;
    rc = stage_input(h, packed_input);
    if (rc) return rc;
    const int p = (int)(h->step_host & 1);
    enqueue_sp_front(h, h->d.input_stage, 1, p, step_wmode(h));
    enqueue_sp_back(h, h->d.input_stage, 1, p, 0, EMIT_ALL, learning && !h->cfg.enable_tm, step_wmode(h));
    h->step_host += 1;
    return HTM_OK;
}

// SpatialPooler.process (networks.py:26-35) one phase per call, for plug-in objects that live on the host: the caller
// (bithtm_amd/networks.py) interleaves these with the `process` / `update` methods of the user's objects.
extern "C" int htm_sp_phase(htm_handle *h, int32_t phase, const void *data, int64_t count) {
    if (!h) return HTM_ERR_ARGUMENT;
    if (h) flush_tail(h);
    REJECT_WHEN_AHEAD(h);
    if (!h->cfg.enable_sp) { h->err = "handle has no Spatial Pooler"; return HTM_ERR_STATE; }
    if (h->world > 1) { h->err = "htm_sp_phase: not available on a column-sharded handle"; return HTM_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    refresh_exchange_mode(h);
    Dev &d = h->d;
    const int p = (int)(h->step_host & 1);
    const int c256 = (d.C + 255) / 256;
    // the top-digit histogram of this step's keys is accumulated by the phase that makes the keys and consumed (and
    // cleared) by the select: a phase that makes keys starts from a clean one, whatever ran before it in this step
    if (phase == HTM_SP_OVERLAP || phase == HTM_SP_BOOST || (phase == HTM_SP_SELECT && data))
        HIPCHK(h, hipMemsetAsync(d.hist0 + (size_t)p * HIST0_PAR, 0, (size_t)HIST0_PAR * 4, h->stream));
    switch (phase) {
        case HTM_SP_OVERLAP: {                     // DenseProjection.process + ExponentialBoosting.process; data = packed input
            if (!data) return HTM_ERR_ARGUMENT;
            int rc = stage_input(h, (const uint32_t *)data);
            if (rc) return rc;
            LAUNCH(h, "sp_overlap", k_sp_overlap, h->sp_blocks, RB, d, d.input_stage, 1, h->G, p, p, 0, 0);
            break;
        }
        case HTM_SP_BOOST: {                       // ExponentialBoosting.process on overlaps from the host; data = int32[C]
            if (!data || count != d.C) { h->err = "htm_sp_phase(BOOST): need column_dim overlaps"; return HTM_ERR_ARGUMENT; }
            HIPCHK(h, hipMemcpyAsync(d.overlap[p], data, (size_t)d.C * 4, hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));         // (the caller's buffer is borrowed for the call only)
            LAUNCH(h, "sp_keys", k_sp_keys, std::min((d.C + RB - 1) / RB, 256), RB, d, p, 1);
            break;
        }
        case HTM_SP_SELECT: {                      // GlobalInhibition.process; data = double[C] boosted overlaps, or NULL: the device's
            if (data) {
                if (count != d.C) { h->err = "htm_sp_phase(SELECT): need column_dim boosted overlaps"; return HTM_ERR_ARGUMENT; }
                HIPCHK(h, hipMemcpyAsync(d.boosted[p], data, (size_t)d.C * 8, hipMemcpyHostToDevice, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
                LAUNCH(h, "sp_keys", k_sp_keys, std::min((d.C + RB - 1) / RB, 256), RB, d, p, 0);
            }
            for (int pass = 1; pass < d.sel_passes; ++pass) LAUNCH(h, "sp_select", k_sel_pass, h->sel_blocks, RB, d, pass, p);
            enqueue_sp_back(h, d.input_stage, 1, p, 0, 0, false);          // the list and its bitmap, nothing else
            break;
        }
        case HTM_SP_ACTIVE: {                      // a winner list from the host; data = int32[count], distinct, any order
            if ((!data && count) || count < 0 || count > d.k) { h->err = "htm_sp_phase(ACTIVE): at most active_columns columns"; return HTM_ERR_ARGUMENT; }
            std::vector<int> cols((const int *)data, (const int *)data + count);
            std::sort(cols.begin(), cols.end());
            for (int64_t i = 0; i < count; ++i)
                if (cols[i] < 0 || cols[i] >= d.C || (i && cols[i] == cols[i - 1])) { h->err = "htm_sp_phase(ACTIVE): bad column list"; return HTM_ERR_ARGUMENT; }
            if (count) HIPCHK(h, hipMemcpyAsync(d.active_cols[p], cols.data(), (size_t)count * 4, hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            h->phase_active = (int)count;
            h->phase_open = true;
            if (d.colbits[p]) {
                LAUNCH(h, "sp_list_bits", k_sp_list_bits, std::max(1, (d.colwords + 255) / 256), 256, d, p, (int)count, 0);
                if (count) LAUNCH(h, "sp_list_bits", k_sp_list_bits, (int)((count + 255) / 256), 256, d, p, (int)count, 1);
            }
            return HTM_OK;
        }
        case HTM_SP_LEARN:                         // DenseProjection.update on the current winner list, with the input of OVERLAP
            if (data) {
                int rc = stage_input(h, (const uint32_t *)data);
                if (rc) return rc;
            }
            if (h->phase_active) LAUNCH(h, "sp_learn", k_sp_learn, h->phase_active, 256, d, d.input_stage, 1, p);
            break;
        case HTM_SP_DUTY:                          // ExponentialBoosting.update on the current winner list
            LAUNCH(h, "sp_duty", k_sp_duty_list, c256, 256, d, p, h->phase_active, 0);
            if (h->phase_active) LAUNCH(h, "sp_duty", k_sp_duty_list, (h->phase_active + 255) / 256, 256, d, p, h->phase_active, 1);
            break;
        case HTM_SP_COMMIT:                        // close the step of a handle without Temporal Memory
            if (h->cfg.enable_tm) { h->err = "htm_sp_phase(COMMIT): the Temporal Memory's step closes the timestep (htm_tm_step)"; return HTM_ERR_STATE; }
            hipLaunchKernelGGL(k_sp_commit, dim3(1), dim3(1), 0, h->stream, d, p);
            h->step_host += 1;
            break;
        default: h->err = "htm_sp_phase: unknown phase"; return HTM_ERR_ARGUMENT;
    }
    if (phase == HTM_SP_SELECT) h->phase_active = d.k;
    h->phase_open = phase != HTM_SP_COMMIT;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("kernel launch: ") + hipGetErrorString(e); return HTM_ERR_HIP; }
    return HTM_OK;
}

extern "C" int htm_tm_step(htm_handle *h, const int32_t *active_column, int32_t n, int32_t learning, int32_t return_winner_cell) {
    if (!h || (!active_column && n > 0)) return HTM_ERR_ARGUMENT;
    if (h) flush_tail(h);
    REJECT_WHEN_AHEAD(h);
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
    LAUNCH(h, "tm_activate", k_tm_activate, std::max(1, (n * d.KP + 255) / 256), 256, d, p, n, want);
    enqueue_tm(h, n, learning ? 1 : 0, want, p, nullptr, 1, false);
    h->step_host += 1;
    h->phase_open = false;
    return HTM_OK;
}

// PredictiveProjection.update (projections.py:257-293) called on its own: learning with the learning cells, the
// punishment mask and the previous State chosen by the caller.  The previous step's side -- prev_state, input_activation,
// winner_input -- is what the handle holds as its previous step (written there with the state import if it is not the
// handle's own: HTM_IMPORT_PREV_STATE).  columns[i] (distinct, any order, at most active_columns) has learning cells
// winner_words[i] (bit j = cell j; `output_learning` of :261-262 for that column) of which unaccounted_words[i] need a new
// segment (:271: learning_output cells whose max jittered potential is below epsilon); punish_words: one word per column
// of the model, bit j = cell j of `output_punishment` (:269), or NULL for "every cell of a column not listed" (what
// TemporalMemory passes, networks.py:107-108,111).  Does not close the timestep: htm_tm_scan does.
extern "C" int htm_tm_update(htm_handle *h, const int32_t *columns, const uint32_t *winner_words, const uint32_t *unaccounted_words,
                             int32_t n, const uint32_t *punish_words) {
    if (!h || (n > 0 && (!columns || !winner_words || !unaccounted_words))) return HTM_ERR_ARGUMENT;
    if (h) flush_tail(h);
    REJECT_WHEN_AHEAD(h);
    if (!h->cfg.enable_tm) { h->err = "handle has no Temporal Memory"; return HTM_ERR_STATE; }
    if (h->world > 1) { h->err = "htm_tm_update: not available on a column-sharded handle"; return HTM_ERR_STATE; }
    Dev &d = h->d;
    if (n < 0 || n > d.k) { h->err = "htm_tm_update: more columns with learning cells than active_columns"; return HTM_ERR_ARGUMENT; }
    // ascending columns (the order of the winner list, networks.py:103-104 under the ascending-column policy)
    std::vector<int> order((size_t)n);
    for (int i = 0; i < n; ++i) order[(size_t)i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return columns[a] < columns[b]; });
    std::vector<int> cols((size_t)n);
    const int WPC = d.WPC;                          // (the words: WPC per listed column)
    std::vector<uint32_t> ww((size_t)n * WPC), uw((size_t)n * WPC);
    for (int i = 0; i < n; ++i) {
        const int o = order[(size_t)i];
        cols[(size_t)i] = columns[o];
        for (int hw = 0; hw < WPC; ++hw) {
            ww[(size_t)i * WPC + hw] = winner_words[(size_t)o * WPC + hw];
            uw[(size_t)i * WPC + hw] = unaccounted_words[(size_t)o * WPC + hw] & winner_words[(size_t)o * WPC + hw];
        }
        if (cols[(size_t)i] < 0 || cols[(size_t)i] >= d.C || (i && cols[(size_t)i] == cols[(size_t)i - 1])) { h->err = "htm_tm_update: bad column list"; return HTM_ERR_ARGUMENT; }
    }
    HIPCHK(h, hipSetDevice(h->device));
    const int p = (int)(h->step_host & 1);
    int *d_cols = nullptr;
    uint32_t *d_ww = nullptr, *d_uw = nullptr, *d_pun = nullptr;
    auto release = [&]() { if (d_cols) hipFree(d_cols); if (d_ww) hipFree(d_ww); if (d_uw) hipFree(d_uw); if (d_pun) hipFree(d_pun); };
    const size_t nb = (size_t)std::max(n, 1) * 4, nbw = nb * WPC, pun_bytes = (size_t)d.C * WPC * 4;
    // punish_words == NULL: every cell of a column that is not listed (networks.py:107-108,111).  The mask is built here: the
    // middle launch's own default reads the step's active words, which this entry point does not write (htm_tm_scan does, later)
    std::vector<uint32_t> default_pun;
    if (!punish_words) {
        default_pun.resize((size_t)d.C * WPC);
        for (size_t w = 0; w < default_pun.size(); ++w) {
            const int cells = d.K - 32 * (int)(w % WPC);
            default_pun[w] = cells >= 32 ? 0xFFFFFFFFu : ((1u << cells) - 1u);
        }
        for (int i = 0; i < n; ++i)
            for (int hw = 0; hw < WPC; ++hw) default_pun[(size_t)cols[(size_t)i] * WPC + hw] = 0u;
        punish_words = default_pun.data();
    }
    if (hipMalloc((void **)&d_cols, nb) != hipSuccess || hipMalloc((void **)&d_ww, nbw) != hipSuccess || hipMalloc((void **)&d_uw, nbw) != hipSuccess ||
        hipMalloc((void **)&d_pun, pun_bytes) != hipSuccess) { release(); h->err = "htm_tm_update: hipMalloc failed"; return HTM_ERR_HIP; }
    bool ok = true;
    if (n) ok = hipMemcpyAsync(d_cols, cols.data(), nb, hipMemcpyHostToDevice, h->stream) == hipSuccess &&
                hipMemcpyAsync(d_ww, ww.data(), nbw, hipMemcpyHostToDevice, h->stream) == hipSuccess &&
                hipMemcpyAsync(d_uw, uw.data(), nbw, hipMemcpyHostToDevice, h->stream) == hipSuccess;
    if (ok) ok = hipMemcpyAsync(d_pun, punish_words, pun_bytes, hipMemcpyHostToDevice, h->stream) == hipSuccess;
    if (!ok) { release(); h->err = "htm_tm_update: hipMemcpy failed"; return HTM_ERR_HIP; }
    hipLaunchKernelGGL(k_tm_ext_winners, dim3(std::min((d.C + 255) / 256, 1024)), dim3(256), 0, h->stream, d, p, d_cols, d_ww, d_uw, n, 0);
    if (n) hipLaunchKernelGGL(k_tm_ext_winners, dim3((n * WPC + 255) / 256), dim3(256), 0, h->stream, d, p, d_cols, d_ww, d_uw, n, 1);
    d.punish = d_pun;                               // (kernels take Dev by value: set for the middle launch only)
    LAUNCH(h, "tm_mid", k_mid_rows, 1 + kClassifyBlocks + h->zero_blocks, 256, d, p, n, 1, 1, kClassifyBlocks, nullptr, 1, 0, 0, 0);
    d.punish = nullptr;
    launch_learn(h, p);
    hipError_t e = hipGetLastError();
    const bool synced = hipStreamSynchronize(h->stream) == hipSuccess;       // (the staging buffers are this call's)
    release();
    if (e != hipSuccess || !synced) { h->err = std::string("htm_tm_update: ") + hipGetErrorString(e != hipSuccess ? e : hipGetLastError()); return HTM_ERR_HIP; }
    h->phase_open = false;
    return HTM_OK;
}

// PredictiveProjection.process (projections.py:245-255) called on its own: the segment scan against the active cells the
// caller names (active_words: one word per column of the model, bit j = cell j), which become the step's cell activation;
// closes the timestep.  The State is read with htm_read (MATCH_*, SEG_POTENTIAL, CELL_MAX_JITTER, CELL_PREDICTION).
extern "C" int htm_tm_scan(htm_handle *h, const uint32_t *active_words) {
    if (!h || !active_words) return HTM_ERR_ARGUMENT;
    if (h) flush_tail(h);
    REJECT_WHEN_AHEAD(h);
    if (!h->cfg.enable_tm) { h->err = "handle has no Temporal Memory"; return HTM_ERR_STATE; }
    if (h->world > 1) { h->err = "htm_tm_scan: not available on a column-sharded handle"; return HTM_ERR_STATE; }
    Dev &d = h->d;
    HIPCHK(h, hipSetDevice(h->device));
    const int p = (int)(h->step_host & 1);
    uint32_t *d_act = nullptr;
    if (hipMalloc((void **)&d_act, (size_t)d.C * d.WPC * 4) != hipSuccess) { h->err = "htm_tm_scan: hipMalloc failed"; return HTM_ERR_HIP; }
    bool ok = hipMemcpyAsync(d_act, active_words, (size_t)d.C * d.WPC * 4, hipMemcpyHostToDevice, h->stream) == hipSuccess;
    // clean accumulators: the scan sets match bits, per-cell maxima and prediction bits with atomics (in a whole timestep the
    // middle launch and the learning role of the step before leave them clean)
    ok = ok && hipMemsetAsync(d.match_bits[p], 0, (size_t)(d.Lcap + 255) / 256 * 8 * 4, h->stream) == hipSuccess &&
         hipMemsetAsync(d.cellmax[p], 0, (size_t)d.C * d.KP * 4, h->stream) == hipSuccess &&
         hipMemsetAsync(&d.ctr->n_active_cells, 0, sizeof(int), h->stream) == hipSuccess;
    if (!ok) { hipFree(d_act); h->err = "htm_tm_scan: staging failed"; return HTM_ERR_HIP; }
    hipLaunchKernelGGL(k_tm_ext_active, dim3(std::min((d.C + 255) / 256, 1024)), dim3(256), 0, h->stream, d, p, d_act);
    launch_scan(h, p, scan_lds(d, 1) <= 64 * 1024);
    hipError_t e = hipGetLastError();
    const bool synced = hipStreamSynchronize(h->stream) == hipSuccess;
    hipFree(d_act);
    if (e != hipSuccess || !synced) { h->err = std::string("htm_tm_scan: ") + hipGetErrorString(e != hipSuccess ? e : hipGetLastError()); return HTM_ERR_HIP; }
    h->step_host += 1;
    h->phase_open = false;
    return HTM_OK;
}

// htm_run, or (dry) only the capture + instantiation of every hipGraph that htm_run call would replay
static int run_or_prepare(htm_handle *h, const uint32_t *device_inputs, int32_t n_inputs, int32_t n_steps, int32_t learning,
                          int32_t use_graph, bool dry) {
    if (h) flush_tail(h);
    if (!h || !device_inputs || n_inputs < 1 || n_steps < 0) return HTM_ERR_ARGUMENT;
    if (!h->cfg.enable_sp || !h->cfg.enable_tm) { h->err = "htm_run needs a handle with SP and TM"; return HTM_ERR_STATE; }
    if (h->world > 1) { h->err = "sharded handle: use htm_shard_begin / htm_shard_finish"; return HTM_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    refresh_exchange_mode(h);
    learning = learning ? 1 : 0;
    // (a short call is launched eagerly whatever the flag says: a graph launch on an idle device starts its first kernel
    // about 7 us later than a kernel launch does, and the host submits three launches per 30-us step with time to spare --
    // measured, 20 steps per call: 615 against 638 us; from 64 steps on the graphs are level and then ahead)
    const bool graph = (use_graph & 1) && !h->profile && n_steps >= h->eager_below;
    const bool pipeline = !(use_graph & 2) && can_pipeline(h);
    const bool resume = sp_is_ahead(h);            // the previous call left the SP one step (and a front) ahead
    if (resume && (h->ahead_bank != device_inputs || h->ahead_n_inputs != n_inputs || h->ahead_learning != learning)) {
        h->err = "htm_run: the previous call ended with HTM_RUN_CONTINUE; this one must use the same bank, n_inputs and learning flag";
        return HTM_ERR_STATE;
    }
    // keep looking ahead past the end of this call -- where the pipelined schedule is available (the flag is a promise of
    // the caller's, not a demand: without the schedule the call simply leaves nothing outstanding)
    const bool cont = (use_graph & 4) && pipeline && n_steps > 0;
    if (dry && !graph) return HTM_OK;
    if (!dry) { int rc = close_open_phases(h); if (rc) return rc; }
    // The SP is ahead but the pipelined schedule is gone (another handle with its own stream has appeared on the device since,
    // or this call asks for HTM_RUN_NO_PIPELINE): the coming step is run as the LAST step of the run that went ahead -- its
    // launches hold no select finish, so nothing in them waits for another block -- and the rest of the call unpipelined.
    if (resume && !pipeline && n_steps > 0) {
        if (dry) return HTM_OK;                      // (that step is launched eagerly; the rest builds its graphs when it runs)
        {
            const int p = (int)(h->step_host & 1);
            const StepPlan last{true, false, false};
            if (h->ahead_lean) enqueue_lean(h, p, learning, device_inputs, n_inputs, last);
            else enqueue_pipelined(h, p, learning, device_inputs, n_inputs, last);
            h->step_host += 1;
            // (the four-launch schedule had begun the step after it: that front is never consumed)
            HIPCHK(h, hipMemsetAsync(h->d.hist0 + (size_t)(h->step_host & 1) * HIST0_PAR, 0, (size_t)HIST0_PAR * 4, h->stream));
            h->ahead_bank = nullptr;
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { h->err = std::string("kernel launch: ") + hipGetErrorString(e); return HTM_ERR_HIP; }
        }
        return run_or_prepare(h, device_inputs, n_inputs, n_steps - 1, learning, use_graph, false);
    }
    // Graphs hold the launches of one step, or of up to kGraphSteps consecutive steady-state steps (a graph
    // launch boundary costs about 5 us more than a kernel boundary inside a graph: tools/step_timeline.py).
    // Nothing in a graph depends on the step index: kernels read it, and with it the bank row, from
    // the device counter.
    const int kGraphSteps = h->graph_steps;
    if (h->seg_pinned) { const int seen = *(volatile int *)h->seg_pinned; h->seg_hint = std::max(h->seg_hint, seen); }      // what the last run left
    bool sp_done = resume;                          // the SP has already done the coming step
    long long step = h->step_host;
    for (int t = 0; t < n_steps;) {
        const bool lean = pipeline && can_lean(h);      // (looks one step ahead, not two)
        const StepPlan plan{sp_done, pipeline && (t + 1 < n_steps || cont), pipeline && !lean && (t + 2 < n_steps || cont)};
        sp_done = plan.next_sp;
        if (!graph) {
            int rc = enqueue_step(h, device_inputs, n_inputs, learning, plan);
            if (rc) return rc;
            t += 1;
            continue;
        }
        const int p = (int)(step & 1);
        // steady state: this and the next span - 1 steps all look ahead fully
        int span = 1;
        if (plan.sp_done && (lean ? plan.next_sp : plan.next_front)) {
            const int steady = cont ? n_steps - t : n_steps - t - (lean ? 1 : 2);       // steps from here on that look ahead fully
            if (cont && steady > 1 && steady < 2 * kGraphSteps) span = steady;      // a continuing call's (last) stretch: one graph
            else if (steady >= kGraphSteps) span = kGraphSteps;
        }
        if (!dry) {
            if (!plan.sp_done && !plan.next_sp) enqueue_sp_front(h, device_inputs, n_inputs, p, step_wmode(h));    // eager
            enqueue_cold_start(h, device_inputs, n_inputs, learning, plan);                         // eager: first step of a pipelined run
        }
        auto key = std::make_tuple(p, learning * 16 + (plan.sp_done ? 4 : 0) + (plan.next_sp ? 2 : 0) + (plan.next_front ? 1 : 0) + 32 * scan_spec_blocks(h) + (scan_pool_is_large(h) ? (1 << 20) : 0) + (h->emit_fused ? (1 << 21) : 0) + (span << 22) + (lean ? 8 : 0) + (step_wmode(h) ? (1 << 19) : 0),
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
        if (!dry) {
            HIPCHK(h, hipGraphLaunch(it->second, h->stream));
            h->step_host += span;
        }
        step += span;
        t += span;
    }
    if (dry) return HTM_OK;
    if (n_steps > 0) {
        if (resume && !cont && n_steps == 1)        // the front computed for the step after this one is never consumed:
            HIPCHK(h, hipMemsetAsync(h->d.hist0 + (size_t)(h->step_host & 1) * HIST0_PAR, 0, (size_t)HIST0_PAR * 4, h->stream));    // its digit histogram
        h->ahead_bank = cont ? device_inputs : nullptr;
        h->ahead_lean = cont && can_lean(h);
        h->ahead_n_inputs = n_inputs;
        h->ahead_learning = learning;
        // leave the segment count where the next call finds it (no wait: it may see the one before)
        if (h->seg_pinned) HIPCHK(h, hipMemcpyAsync(h->seg_pinned, &h->d.ctr->S, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    }
    return HTM_OK;
}

extern "C" int htm_run(htm_handle *h, const uint32_t *device_inputs, int32_t n_inputs, int32_t n_steps, int32_t learning, int32_t use_graph) {
    return run_or_prepare(h, device_inputs, n_inputs, n_steps, learning, use_graph, false);
}

extern "C" int htm_prepare(htm_handle *h, const uint32_t *device_inputs, int32_t n_inputs, int32_t n_steps, int32_t learning, int32_t use_graph) {
    return run_or_prepare(h, device_inputs, n_inputs, n_steps, learning, use_graph, true);
}

extern "C" int htm_run_plan(htm_handle *h, int32_t n_steps, int32_t use_graph) {
    if (!h || n_steps < 0) return HTM_ERR_ARGUMENT;
    if (!h->cfg.enable_sp || !h->cfg.enable_tm || h->world > 1) { h->err = "htm_run_plan: htm_run needs an unsharded handle with SP and TM"; return HTM_ERR_STATE; }
    refresh_exchange_mode(h);
    if (h->seg_pinned) { const int seen = *(volatile int *)h->seg_pinned; h->seg_hint = std::max(h->seg_hint, seen); }
    const bool graph = (use_graph & 1) && !h->profile && n_steps >= h->eager_below;
    const bool pipeline = !(use_graph & 2) && can_pipeline(h) && n_steps > 1;
    return (graph ? HTM_PLAN_GRAPH : 0) | (pipeline ? HTM_PLAN_PIPELINED : 0) | (pipeline && can_lean(h) ? HTM_PLAN_LEAN : 0) |
           (scan_pool_is_large(h) ? HTM_PLAN_SCAN_LARGE : 0);
}

static int read_counters(htm_handle *h, Counters *out);
static int stage_input(htm_handle *h, const uint32_t *packed_input);
static int ensure_shard_buffers(htm_handle *h);

extern "C" int64_t htm_shard_record_bytes(htm_handle *h) {
    if (!h) return HTM_ERR_ARGUMENT;
    return (int64_t)shard_record_bytes(h->d.cand_cap);
}

// front_done: this step's overlap (own columns) was computed beside the previous step's learning and scan (htm_shard_run)
static int shard_enqueue_begin(htm_handle *h, const uint32_t *bank, int n_inputs, void *send_device, bool front_done = false) {
    Dev &d = h->d;
    const int p = (int)(h->step_host & 1);
    d.send = (unsigned char *)send_device;
    // own columns: overlap + boost + histogram (unless computed ahead); the local select's finish, the cell words each
    // candidate would have if it became active, the record -- and, in further blocks of that launch, the zeroing of the step's
    // dense words
    // (the local select: one windowed histogram pass beside the overlap, finished inside the candidates kernel;
    // BITHTM_SHARD_WINDOW=0: two launched digits.  While another handle with a stream of its own is live on the device the
    // blocks of a grid must not wait for each other: every digit by a launch, the counts by k_sp_count -- same candidates)
    const int fused = h->emit_fused ? 1 : 0;
    const int wmode = fused ? h->knob_shard_window : 0;
    if (front_done && h->shard_front_wmode != wmode) {       // the exchange mode changed since the front was computed: start over
        HIPCHK(h, hipMemsetAsync(d.hist0 + (size_t)p * HIST0_PAR, 0, (size_t)HIST0_PAR * 4, h->stream));
        front_done = false;
    }
    if (!front_done) LAUNCH(h, "shard_overlap", k_shard_overlap, h->sp_blocks, RB, d, bank, n_inputs, h->G, p, wmode, 0);
    if (!wmode)
        for (int pass = 1; pass < d.sel_passes; ++pass) LAUNCH(h, "sp_select", k_sel_pass, h->sel_blocks, RB, d, pass, p);
    if (!fused) LAUNCH(h, "sp_count", k_sp_count, h->c256_blocks, 256, d, p);
    LAUNCH(h, "shard_candidates", k_sp_emit, h->c256_blocks + std::min((d.C + 255) / 256, 64), 256, d, p, 1, fused, EMIT_LOCAL, wmode, h->c256_blocks);
    return 0;
}

// front_next: compute the coming step's overlap beside this step's learning and scan (same bank: htm_shard_run)
static int shard_enqueue_finish(htm_handle *h, const uint32_t *bank, int n_inputs, const void *recv_device, int learning, bool front_next = false) {
    Dev &d = h->d;
    const int p = (int)(h->step_host & 1);
    LAUNCH(h, "shard_select", k_shard_select, h->world + 1, 1024, d, (const unsigned char *)recv_device, p);      // (+ 1: the death reports)
    const int wmode = h->emit_fused ? h->knob_shard_window : 0;
    enqueue_tm(h, d.k, learning, 1, p, bank, n_inputs, true, front_next ? wmode : -1);
    if (front_next) h->shard_front_wmode = wmode;
    h->step_host += 1;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("kernel launch: ") + hipGetErrorString(e); return HTM_ERR_HIP; }
    return 0;
}

extern "C" int htm_shard_begin(htm_handle *h, const uint32_t *device_inputs, int32_t n_inputs, const uint32_t *packed_input,
                               int32_t learning, void *send_device) {
    if (!h || !send_device || (!device_inputs == !packed_input)) return HTM_ERR_ARGUMENT;
    if (h->world < 2) { h->err = "htm_shard_begin: handle is not sharded"; return HTM_ERR_STATE; }
    if (h->shard_open) { h->err = "htm_shard_begin: previous step not finished"; return HTM_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    refresh_exchange_mode(h);
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
    int rc = shard_enqueue_begin(h, h->shard_bank, h->shard_n_inputs, send_device);
    if (rc) return rc;
    h->shard_open = true;
    (void)learning;
    return HTM_OK;
}

extern "C" int htm_shard_finish(htm_handle *h, const void *recv_device, int32_t learning) {
    if (!h || !recv_device) return HTM_ERR_ARGUMENT;
    if (h->world < 2 || !h->shard_open) { h->err = "htm_shard_finish: no step in progress"; return HTM_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    h->shard_open = false;
    return shard_enqueue_finish(h, h->shard_bank, h->shard_n_inputs, recv_device, learning ? 1 : 0);
}

// ---- the exchange inside the library: RCCL, loaded at run time (a handle that is never sharded needs no RCCL) ----
namespace {
struct RcclApi {
    void *lib = nullptr;
    int (*get_unique_id)(void *) = nullptr;
    int (*comm_init_rank)(void **, int, ncclUniqueIdBytes, int) = nullptr;
    int (*all_gather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*comm_destroy)(void *) = nullptr;
    int (*comm_count)(void *, int *) = nullptr;
    const char *(*get_error_string)(int) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;

const char *load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return nullptr;
    void *lib = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) return "cannot load librccl.so";
    RcclApi api;
    api.lib = lib;
    api.get_unique_id = (int (*)(void *))dlsym(lib, "ncclGetUniqueId");
    api.comm_init_rank = (int (*)(void **, int, ncclUniqueIdBytes, int))dlsym(lib, "ncclCommInitRank");
    api.all_gather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(lib, "ncclAllGather");
    api.comm_destroy = (int (*)(void *))dlsym(lib, "ncclCommDestroy");
    api.get_error_string = (const char *(*)(int))dlsym(lib, "ncclGetErrorString");
    api.comm_count = (int (*)(void *, int *))dlsym(lib, "ncclCommCount");
    if (!api.get_unique_id || !api.comm_init_rank || !api.all_gather || !api.comm_destroy) return "librccl.so lacks an expected symbol";
    g_rccl = api;
    g_rccl_destroy = api.comm_destroy;
    return nullptr;
}
}  // namespace

extern "C" int htm_shard_unique_id(void *out128) {
    if (!out128) return HTM_ERR_ARGUMENT;
    if (const char *err = load_rccl()) { g_create_error = err; return HTM_ERR_HIP; }
    ncclUniqueIdBytes id;
    if (g_rccl.get_unique_id(&id) != 0) { g_create_error = "ncclGetUniqueId failed"; return HTM_ERR_HIP; }
    memcpy(out128, &id, sizeof(id));
    return HTM_OK;
}

extern "C" int htm_keyed_draws(uint32_t seed, int32_t stream, uint32_t step, const uint32_t *a, const uint32_t *b, int64_t n, double *out) {
    if (stream < 1 || stream > 5 || n < 0 || (n && (!a || !out))) return HTM_ERR_ARGUMENT;
    const uint32_t base = htm_stream_base(seed, (uint32_t)stream, step);
    for (int64_t i = 0; i < n; ++i) out[i] = (double)htm_draw24(base, a[i], b ? b[i] : 0u) * (1.0 / 16777216.0);
    return HTM_OK;
}

// RCCL round trip at world size 1 on `device`: load the library, create a communicator, all-gather a buffer on a
// stream, compare, destroy.  What a one-GPU box can verify of the in-library exchange (the symbols, the by-value
// unique id, the call on a non-default stream); returns 0 or a negative status (htm_last_error(NULL)).
extern "C" int htm_rccl_selftest(int32_t device) {
    if (const char *err = load_rccl()) { g_create_error = err; return HTM_ERR_HIP; }
    if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return HTM_ERR_HIP; }
    ncclUniqueIdBytes id;
    if (g_rccl.get_unique_id(&id) != 0) { g_create_error = "ncclGetUniqueId failed"; return HTM_ERR_HIP; }
    void *comm = nullptr;
    if (g_rccl.comm_init_rank(&comm, 1, id, 0) != 0) { g_create_error = "ncclCommInitRank failed"; return HTM_ERR_HIP; }
    hipStream_t stream;
    unsigned char *a = nullptr, *b = nullptr;
    const size_t n = 27264;                         // the record of 1311 candidates
    std::vector<unsigned char> src(n), dst(n, 0);
    for (size_t i = 0; i < n; ++i) src[i] = (unsigned char)(i * 131u + 7u);
    bool ok = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess && hipMalloc((void **)&a, n) == hipSuccess &&
              hipMalloc((void **)&b, n) == hipSuccess && hipMemcpy(a, src.data(), n, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && g_rccl.all_gather(a, b, n, 0, comm, stream) == 0 && hipStreamSynchronize(stream) == hipSuccess &&
         hipMemcpy(dst.data(), b, n, hipMemcpyDeviceToHost) == hipSuccess && dst == src;
    // and the same collective captured into a hipGraph and replayed (what htm_shard_run does with whole timesteps)
    int captured = 0;
    if (ok) {
        hipGraph_t graph_obj = nullptr;
        hipGraphExec_t exec = nullptr;
        std::fill(dst.begin(), dst.end(), 0);
        bool cap = hipMemset(b, 0, n) == hipSuccess && hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (cap) {
            const bool in = g_rccl.all_gather(a, b, n, 0, comm, stream) == 0;
            cap = hipStreamEndCapture(stream, &graph_obj) == hipSuccess && in && graph_obj &&
                  hipGraphInstantiate(&exec, graph_obj, nullptr, nullptr, 0) == hipSuccess;
        }
        if (cap) cap = hipGraphLaunch(exec, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess &&
                       hipMemcpy(dst.data(), b, n, hipMemcpyDeviceToHost) == hipSuccess && dst == src;
        if (exec) hipGraphExecDestroy(exec);
        if (graph_obj) hipGraphDestroy(graph_obj);
        (void)hipGetLastError();
        captured = cap ? 1 : 0;
    }
    g_rccl.comm_destroy(comm);
    if (a) hipFree(a);
    if (b) hipFree(b);
    hipStreamDestroy(stream);
    if (!ok) { g_create_error = "RCCL all-gather self-test failed"; return HTM_ERR_HIP; }
    return captured ? HTM_OK : 1;                   // 1: the collective works but this runtime does not capture it (htm_shard_run then launches eagerly)
}

extern "C" int htm_shard_comm_init(htm_handle *h, const void *unique_id128) {
    if (!h || !unique_id128) return HTM_ERR_ARGUMENT;
    if (h->world < 2) { h->err = "htm_shard_comm_init: handle is not sharded"; return HTM_ERR_STATE; }
    if (h->rccl_comm) { h->err = "htm_shard_comm_init: already initialised"; return HTM_ERR_STATE; }
    if (const char *err = load_rccl()) { h->err = err; return HTM_ERR_HIP; }
    HIPCHK(h, hipSetDevice(h->device));
    ncclUniqueIdBytes id;
    memcpy(&id, unique_id128, sizeof(id));
    void *comm = nullptr;
    const int rc = g_rccl.comm_init_rank(&comm, h->world, id, h->rank);
    if (rc != 0) { h->err = std::string("ncclCommInitRank: ") + (g_rccl.get_error_string ? g_rccl.get_error_string(rc) : "failed"); return HTM_ERR_HIP; }
    h->rccl_comm = comm;
    int rc2 = ensure_shard_buffers(h);
    if (rc2) return rc2;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    // Preflight on THIS communicator and THESE buffers, collectively (every rank runs the same two gathers): the exchange of a
    // timestep launched eagerly -- its result checked: rank q's part of the gathered buffer must hold q's pattern -- and the same
    // collective captured into a hipGraph and replayed.  htm_shard_run replays whole timesteps as graphs only if the second
    // check passes here (a rank that launches eagerly and one that replays issue the same collective: the modes may differ).
    {
        const size_t rb = shard_record_bytes(h->d.cand_cap);
        std::vector<unsigned char> got(rb * (size_t)h->world);
        auto check = [&]() -> bool {
            if (hipStreamSynchronize(h->stream) != hipSuccess || hipMemcpy(got.data(), h->shard_recv, got.size(), hipMemcpyDeviceToHost) != hipSuccess) return false;
            for (int q = 0; q < h->world; ++q)
                for (size_t i = 0; i < rb; i += 997)
                    if (got[(size_t)q * rb + i] != (unsigned char)(q + 1)) return false;
            return true;
        };
        HIPCHK(h, hipMemsetAsync(h->shard_send, h->rank + 1, rb, h->stream));
        HIPCHK(h, hipMemsetAsync(h->shard_recv, 0, rb * (size_t)h->world, h->stream));
        // (a rank whose first gather fails still issues the second: its peers are on their way into theirs and would wait for
        // it for ever; the failure is reported after both)
        int nrc = g_rccl.all_gather(h->shard_send, h->shard_recv, rb, 0, comm, h->stream);
        const bool first_ok = nrc == 0 && check();
        bool ok = first_ok && h->stream != nullptr && hipMemsetAsync(h->shard_recv, 0, rb * (size_t)h->world, h->stream) == hipSuccess &&
                  hipStreamSynchronize(h->stream) == hipSuccess;      // (the default stream cannot be captured)
        hipGraph_t graph_obj = nullptr;
        hipGraphExec_t exec = nullptr;
        if (ok && hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const bool in = g_rccl.all_gather(h->shard_send, h->shard_recv, rb, 0, comm, h->stream) == 0;
            ok = hipStreamEndCapture(h->stream, &graph_obj) == hipSuccess && in && graph_obj && hipGraphInstantiate(&exec, graph_obj, nullptr, nullptr, 0) == hipSuccess;
        } else {
            ok = false;
        }
        // (every rank must issue the second gather, captured or not: the others are waiting in theirs)
        if (ok) ok = hipGraphLaunch(exec, h->stream) == hipSuccess;
        else ok = g_rccl.all_gather(h->shard_send, h->shard_recv, rb, 0, comm, h->stream) == 0 && false;
        const bool delivered = check();
        if (exec) hipGraphExecDestroy(exec);
        if (graph_obj) hipGraphDestroy(graph_obj);
        (void)hipGetLastError();
        h->shard_graph_ok = ok && delivered;
        HIPCHK(h, hipMemsetAsync(h->shard_send, 0, rb, h->stream));
        HIPCHK(h, hipMemsetAsync(h->shard_recv, 0, rb * (size_t)h->world, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (!first_ok) { h->err = "htm_shard_comm_init: the all-gather over this communicator does not deliver the ranks' records"; return HTM_ERR_HIP; }
    }
    return HTM_OK;
}

// 1: htm_shard_run replays whole timesteps (the all-gather included) as hipGraphs on this handle; 0: it launches eagerly
// (the preflight of htm_shard_comm_init found the collective not capturable, or the handle enqueues on the default stream)
extern "C" int htm_shard_graph_ok(htm_handle *h) {
    if (!h) return HTM_ERR_ARGUMENT;
    return h->shard_graph_ok ? 1 : 0;
}

// ranks of the RCCL communicator htm_shard_step / htm_shard_run exchange over (ncclCommCount): what a scaling run reports
extern "C" int htm_shard_comm_size(htm_handle *h) {
    if (!h) return HTM_ERR_ARGUMENT;
    if (!h->rccl_comm || !g_rccl.comm_count) { h->err = "htm_shard_comm_size: no communicator (htm_shard_comm_init)"; return HTM_ERR_STATE; }
    int n = 0;
    if (g_rccl.comm_count(h->rccl_comm, &n) != 0) { h->err = "ncclCommCount failed"; return HTM_ERR_HIP; }
    return n;
}

static int ensure_shard_buffers(htm_handle *h) {
    if (h->shard_send) return 0;
    const size_t rb = shard_record_bytes(h->d.cand_cap);
    int rc = dalloc(h, &h->shard_send, rb);
    rc |= dalloc(h, &h->shard_recv, rb * (size_t)h->world);
    return rc;
}

// All the shards of a model inside ONE process on one device (tests, single-GPU rehearsals, bench.py's configs[4]
// leg): handles[r] is rank r of n = shard_world, all of them on the same stream; the all-gather is n x n device copies.
extern "C" int htm_shard_group_step(htm_handle *const *handles, int32_t n, const uint32_t *const *device_inputs, int32_t n_inputs,
                                    const uint32_t *packed_input, int32_t learning) {
    if (!handles || n < 2 || (!device_inputs == !packed_input)) return HTM_ERR_ARGUMENT;
    for (int r = 0; r < n; ++r) {
        htm_handle *h = handles[r];
        if (!h || h->world != n || h->rank != r || h->stream != handles[0]->stream || h->device != handles[0]->device) {
            if (h) h->err = "htm_shard_group_step: handles must be ranks 0..n-1 of one group on one stream";
            return HTM_ERR_ARGUMENT;
        }
    }
    for (int r = 0; r < n; ++r) {
        htm_handle *h = handles[r];
        HIPCHK(h, hipSetDevice(h->device));
        refresh_exchange_mode(h);
        int rc = ensure_shard_buffers(h);
        if (rc) return rc;
        const uint32_t *bank = device_inputs ? device_inputs[r] : nullptr;
        if (packed_input) {
            rc = stage_input(h, packed_input);
            if (rc) return rc;
            bank = h->d.input_stage;
        }
        h->shard_bank = bank;
        h->shard_n_inputs = packed_input ? 1 : n_inputs;
        rc = shard_enqueue_begin(h, h->shard_bank, h->shard_n_inputs, h->shard_send);
        if (rc) return rc;
    }
    const size_t rb = shard_record_bytes(handles[0]->d.cand_cap);
    for (int r = 0; r < n; ++r)
        for (int q = 0; q < n; ++q)
            HIPCHK(handles[r], hipMemcpyAsync(handles[r]->shard_recv + (size_t)q * rb, handles[q]->shard_send, rb, hipMemcpyDeviceToDevice, handles[r]->stream));
    for (int r = 0; r < n; ++r) {
        int rc = shard_enqueue_finish(handles[r], handles[r]->shard_bank, handles[r]->shard_n_inputs, handles[r]->shard_recv, learning ? 1 : 0);
        if (rc) return rc;
    }
    return HTM_OK;
}

extern "C" int htm_shard_step(htm_handle *h, const uint32_t *device_inputs, int32_t n_inputs, const uint32_t *packed_input, int32_t learning) {
    if (!h || (!device_inputs == !packed_input)) return HTM_ERR_ARGUMENT;
    if (h->world < 2 || !h->rccl_comm) { h->err = "htm_shard_step: needs a sharded handle after htm_shard_comm_init"; return HTM_ERR_STATE; }
    if (h->shard_open) { h->err = "htm_shard_step: a step opened with htm_shard_begin is not finished"; return HTM_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    refresh_exchange_mode(h);
    const uint32_t *bank = device_inputs;
    if (packed_input) {
        int rc = stage_input(h, packed_input);
        if (rc) return rc;
        bank = h->d.input_stage;
        n_inputs = 1;
    } else if (n_inputs < 1) {
        return HTM_ERR_ARGUMENT;
    }
    int rc = shard_enqueue_begin(h, bank, n_inputs, h->shard_send);
    if (rc) return rc;
    const size_t rb = shard_record_bytes(h->d.cand_cap);
    const int nrc = g_rccl.all_gather(h->shard_send, h->shard_recv, rb, /* ncclChar */ 0, h->rccl_comm, h->stream);
    if (nrc != 0) { h->err = std::string("ncclAllGather: ") + (g_rccl.get_error_string ? g_rccl.get_error_string(nrc) : "failed"); return HTM_ERR_HIP; }
    return shard_enqueue_finish(h, bank, n_inputs, h->shard_recv, learning ? 1 : 0);
}

// ---- n timesteps of the column-sharded step without the host in the loop (htm_shard_run, htm_shard_group_run) ----
// Inside a run the overlap of step t + 1 (own columns) rides in the last launch of step t, and the launches of whole
// steps -- the collective included: RCCL's all-gather is captured like a kernel -- are replayed as hipGraphs (two parities
// per graph, up to 16 steps).  hs = the handles stepped together (one: this process's rank, the exchange by RCCL;
// several: all ranks of a group in one process, the exchange as device copies).
static int shard_exchange(htm_handle *const *hs, int n) {
    if (n == 1) {
        htm_handle *h = hs[0];
        const int nrc = g_rccl.all_gather(h->shard_send, h->shard_recv, shard_record_bytes(h->d.cand_cap), /* ncclChar */ 0, h->rccl_comm, h->stream);
        if (nrc != 0) { h->err = std::string("ncclAllGather: ") + (g_rccl.get_error_string ? g_rccl.get_error_string(nrc) : "failed"); return HTM_ERR_HIP; }
        return 0;
    }
    const size_t rb = shard_record_bytes(hs[0]->d.cand_cap);
    for (int r = 0; r < n; ++r)
        for (int q = 0; q < n; ++q)
            HIPCHK(hs[r], hipMemcpyAsync(hs[r]->shard_recv + (size_t)q * rb, hs[q]->shard_send, rb, hipMemcpyDeviceToDevice, hs[r]->stream));
    return 0;
}

// `count` consecutive steps; first_front_done: the first one's overlap exists already; last_front_next: the last one
// computes the overlap of the step after it
static int shard_enqueue_steps(htm_handle *const *hs, int n, const uint32_t *const *banks, int n_inputs, int learning, int count,
                               bool first_front_done, bool last_front_next) {
    for (int i = 0; i < count; ++i) {
        const bool fd = i > 0 || first_front_done, fn = i + 1 < count || last_front_next;
        for (int r = 0; r < n; ++r) {
            int rc = shard_enqueue_begin(hs[r], banks[r], n_inputs, hs[r]->shard_send, fd);
            if (rc) return rc;
        }
        int rc = shard_exchange(hs, n);
        if (rc) return rc;
        for (int r = 0; r < n; ++r) {
            rc = shard_enqueue_finish(hs[r], banks[r], n_inputs, hs[r]->shard_recv, learning, fn);
            if (rc) return rc;
        }
    }
    return 0;
}

static int shard_run(htm_handle *const *hs, int n, const uint32_t *const *banks, int n_inputs, int n_steps, int learning, int use_graph) {
    htm_handle *h0 = hs[0];
    learning = learning ? 1 : 0;
    bool pipeline = !(use_graph & 2), graph = (use_graph & 1) != 0;
    for (int r = 0; r < n; ++r) {
        htm_handle *h = hs[r];
        HIPCHK(h, hipSetDevice(h->device));
        refresh_exchange_mode(h);
        int rc = ensure_shard_buffers(h);
        if (rc) return rc;
        if (h->seg_pinned) { const int seen = *(volatile int *)h->seg_pinned; h->seg_hint = std::max(h->seg_hint, seen); }
        if (h->profile) graph = false;
    }
    if (n == 1 && !h0->shard_graph_ok) graph = false;      // (this communicator's collective does not replay from a graph: the preflight said so)
    const int kSpan = 16;
    for (int t = 0; t < n_steps;) {
        // steady state: steps whose overlap was computed ahead and that compute the next one's; the first and the last step
        // of a call are launched on their own
        const bool fd = pipeline && t > 0, steady = fd && t + 1 < n_steps;
        int span = 1;
        if (steady && graph) span = std::min(kSpan, (n_steps - 1 - t) & ~1) > 0 ? std::min(kSpan, (n_steps - 1 - t) & ~1) : 1;
        const bool fn = pipeline && t + span < n_steps;
        if (!graph) {
            int rc = shard_enqueue_steps(hs, n, banks, n_inputs, learning, span, fd, fn);
            if (rc) return rc;
            t += span;
            continue;
        }
        std::vector<int> params{(int)(h0->step_host & 1) | learning << 1 | (fd ? 4 : 0) | (fn ? 8 : 0) | span << 4, n, n_inputs};
        for (int r = 0; r < n; ++r) {         // what the launches of a rank depend on besides its arguments
            params.push_back(scan_spec_blocks(hs[r]));
            params.push_back((scan_pool_is_large(hs[r]) ? 1 : 0) | (hs[r]->emit_fused ? 2 : 0));
        }
        auto key = std::make_pair(params, (const void *)banks[0]);
        auto it = h0->shard_graphs.find(key);
        if (it == h0->shard_graphs.end()) {
            std::vector<long long> saved((size_t)n);
            for (int r = 0; r < n; ++r) saved[(size_t)r] = hs[r]->step_host;
            hipGraph_t graph_obj = nullptr;
            hipError_t e = hipStreamBeginCapture(h0->stream, hipStreamCaptureModeThreadLocal);
            int rc = e == hipSuccess ? shard_enqueue_steps(hs, n, banks, n_inputs, learning, span, fd, fn) : HTM_ERR_HIP;
            hipError_t e2 = e == hipSuccess ? hipStreamEndCapture(h0->stream, &graph_obj) : e;
            for (int r = 0; r < n; ++r) hs[r]->step_host = saved[(size_t)r];      // (captured, not run)
            hipGraphExec_t exec = nullptr;
            if (rc == 0 && e2 == hipSuccess && graph_obj && hipGraphInstantiate(&exec, graph_obj, nullptr, nullptr, 0) != hipSuccess) exec = nullptr;
            if (graph_obj) hipGraphDestroy(graph_obj);
            if (!exec) {                            // (a collective that cannot be captured on this runtime: launch eagerly from here on)
                (void)hipGetLastError();
                graph = false;
                h0->err.clear();
                continue;
            }
            it = h0->shard_graphs.emplace(key, exec).first;
        }
        HIPCHK(h0, hipGraphLaunch(it->second, h0->stream));
        for (int r = 0; r < n; ++r) hs[r]->step_host += span;
        t += span;
    }
    if (n_steps > 0)
        for (int r = 0; r < n; ++r)
            if (hs[r]->seg_pinned) HIPCHK(hs[r], hipMemcpyAsync(hs[r]->seg_pinned, &hs[r]->d.ctr->L, sizeof(int), hipMemcpyDeviceToHost, hs[r]->stream));
    return HTM_OK;
}

extern "C" int htm_shard_run(htm_handle *h, const uint32_t *device_inputs, int32_t n_inputs, int32_t n_steps, int32_t learning, int32_t use_graph) {
    if (!h || !device_inputs || n_inputs < 1 || n_steps < 0) return HTM_ERR_ARGUMENT;
    if (h->world < 2 || !h->rccl_comm) { h->err = "htm_shard_run: needs a sharded handle after htm_shard_comm_init"; return HTM_ERR_STATE; }
    if (h->shard_open) { h->err = "htm_shard_run: a step opened with htm_shard_begin is not finished"; return HTM_ERR_STATE; }
    htm_handle *hs[1] = {h};
    const uint32_t *banks[1] = {device_inputs};
    return shard_run(hs, 1, banks, n_inputs, n_steps, learning, use_graph);
}

extern "C" int htm_shard_group_run(htm_handle *const *handles, int32_t n, const uint32_t *const *device_inputs, int32_t n_inputs, int32_t n_steps,
                                   int32_t learning, int32_t use_graph) {
    if (!handles || n < 2 || !device_inputs || n_inputs < 1 || n_steps < 0) return HTM_ERR_ARGUMENT;
    for (int r = 0; r < n; ++r) {
        htm_handle *h = handles[r];
        if (!h || h->world != n || h->rank != r || h->stream != handles[0]->stream || h->device != handles[0]->device || !device_inputs[r]) {
            if (h) h->err = "htm_shard_group_run: handles must be ranks 0..n-1 of one group on one stream, each with its bank";
            return HTM_ERR_ARGUMENT;
        }
    }
    return shard_run(handles, n, device_inputs, n_inputs, n_steps, learning, use_graph);
}

// Pre-populated pool (BASELINE.json configs[4]: a pure scan stress, not a learned state): every cell with flat id in
// [cell_begin, cell_end) gets segments_per_cell segments of `synapses` synapses to keyed-random presynaptic cells
// (no two alike within a segment), permanences keyed-uniform in [perm_lo, perm_hi).  Segment ids are cell-major:
// (cell - cell_begin) * segments_per_cell + j.  Generated on the device (oracle twin: TemporalMemoryOracle.populate);
// a column-sharded handle generates the rows of its own cells only, every handle of the group is given the same range.
extern "C" int htm_populate(htm_handle *h, int64_t cell_begin, int64_t cell_end, int32_t segments_per_cell, int32_t synapses,
                            double perm_lo, double perm_hi, uint32_t seed) {
    if (!h) return HTM_ERR_ARGUMENT;
    if (h) flush_tail(h);
    if (!h->cfg.enable_tm) { h->err = "handle has no Temporal Memory"; return HTM_ERR_STATE; }
    Dev &d = h->d;
    const int64_t N = (int64_t)d.C * d.K;
    if (cell_begin < 0 || cell_end > N || cell_begin > cell_end || segments_per_cell < 1 || synapses < d.match_thr || synapses > 64 ||
        synapses > d.E || synapses > N || !(perm_lo >= 0.0) || !(perm_hi >= perm_lo)) { h->err = "htm_populate: bad arguments"; return HTM_ERR_ARGUMENT; }
    Counters c;
    int rc = read_counters(h, &c);
    if (rc) return rc;
    if (c.S != 0 || h->step_host != 0) { h->err = "htm_populate: the handle must be fresh"; return HTM_ERR_STATE; }
    const int64_t total = (cell_end - cell_begin) * segments_per_cell;
    const int64_t own_lo = std::max<int64_t>(cell_begin, (int64_t)d.c0 * d.K), own_hi = std::min<int64_t>(cell_end, (int64_t)d.c1 * d.K);
    const int64_t own = std::max<int64_t>(own_hi - own_lo, 0) * segments_per_cell;
    if (total > d.Scap || own > d.Lcap) { h->err = "htm_populate: pool too small (segment_capacity / segment_capacity_local)"; return HTM_ERR_CAPACITY; }
    if (own > 0) {
        const int64_t blocks = std::min<int64_t>((own + 3) / 4, (int64_t)h->cus * 64);
        hipLaunchKernelGGL(k_tm_populate, dim3((unsigned)blocks), dim3(256), 0, h->stream, d, (long long)cell_begin, (long long)own_lo, (long long)own,
                           segments_per_cell, synapses, perm_lo, perm_hi, seed);
    }
    c.S = (int32_t)total;
    c.L = (int32_t)own;
    HIPCHK(h, hipMemcpyAsync(&d.ctr->S, &c.S, sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(&d.ctr->L, &c.L, sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->seg_hint = (int)(h->world > 1 ? own : total);
    if (h->seg_pinned) *h->seg_pinned = h->seg_hint;
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
    h->seg_hint = h->world > 1 ? out->L : out->S;   // exact: the stream is idle (rows the scan covers)
    if (h->seg_pinned) *h->seg_pinned = h->seg_hint;
    return 0;
}

extern "C" int htm_get_info(htm_handle *h, htm_info *out) {
    if (!h || !out) return HTM_ERR_ARGUMENT;
    if (h) flush_tail(h);
    Counters c;
    int rc = read_counters(h, &c);
    if (rc) return rc;
    const int q = (int)((h->step_host + 1) & 1);          // parity of the last completed step
    const int rows = h->world > 1 ? c.L : c.S;
    out->step_index = h->step_host;
    out->segments = c.S;
    out->local_segments = rows;
    out->matching_segments = 0;
    if (c.has_distal && rows > 0) {
        std::vector<uint32_t> bits(((size_t)rows + 31) / 32);
        HIPCHK(h, hipMemcpy(bits.data(), h->d.match_bits[q], bits.size() * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < rows; ++i) out->matching_segments += (bits[(size_t)i >> 5] >> (i & 31)) & 1u;
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
    out->candidate_exact_steps = c.cand_exact;
    out->hot_select_steps = c.hot_selects;
    out->select_zoom_steps = c.sel_zooms;
    if (c.error) {
        h->err = std::string("capacity exhausted:") + ((c.error & 1) ? " segment pool (segment_capacity)" : "") +
                 ((c.error & 2) ? " synapse slots (segment_slots)" : "") + ((c.error & 4) ? " work list / growth staging" : "") +
                 ((c.error & 8) ? " dead-segment report (DEAD_CAP)" : "") +
                 ((c.error & 16) ? " (internal) block hand-off timed out in k_sp_emit" : "") +
                 ((c.error & 32) ? " (internal) the middle role's wait for the activation blocks timed out in k_act_mid_rows" : "");
        return HTM_ERR_CAPACITY;          // *out is filled in all the same
    }
    return HTM_OK;
}

// conversions between the internal cell encoding (col * KP + cell: 32 cell slots per column, 64 for cell_dim above 32) and the
// ABI's flat ids
static inline int enc_flat(int enc, int K) { return K > 32 ? (enc >> 6) * K + (enc & 63) : (enc >> 5) * K + (enc & 31); }
static inline int flat_enc(int flat, int K) { return (flat / K) * (K > 32 ? 64 : 32) + (flat % K); }

extern "C" int64_t htm_read(htm_handle *h, int32_t field, void *dst, int64_t count) {
    if (!h || !dst) return HTM_ERR_ARGUMENT;
    if (h) flush_tail(h);
    // the fields whose length the handle's shape gives need no counters: the stream is waited for, nothing else (a State of the
    // reference's API is a dozen reads, and the counters' copy was half of each small one)
    const bool fixed_size = field == HTM_F_ACTIVE_COLUMN || field == HTM_F_OVERLAPS || field == HTM_F_BOOSTED || field == HTM_F_DUTY_CYCLE ||
                            field == HTM_F_CELL_ACTIVATION || field == HTM_F_CELL_PREDICTION || field == HTM_F_WINNER_WORDS ||
                            field == HTM_F_BURSTING || field == HTM_F_SEGCOUNT || field == HTM_F_CELL_MAX_JITTER;
    Counters c;
    memset(&c, 0, sizeof(c));
    int rc = 0;
    if (fixed_size) {
        HIPCHK(h, hipSetDevice(h->device));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    } else {
        rc = read_counters(h, &c);
        if (rc) return rc;
    }
    Dev &d = h->d;
    const int q = (int)((h->step_host + 1) & 1);
    const int qs = h->phase_open ? (int)(h->step_host & 1) : q;      // Spatial Pooler fields while a step is run phase by phase
    const bool sp = h->cfg.enable_sp, tm = h->cfg.enable_tm;
    const int64_t C = d.C, K = d.K, S = h->world > 1 ? c.L : c.S, E = d.E, CW = (int64_t)d.C * d.WPC, KP = d.KP;      // S: rows of the per-segment fields; CW: cell words
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
        case HTM_F_ACTIVE_COLUMN: if ((n = need(true, d.k)) < 0) return n; return copy(d.active_cols[qs], n, 4);
        case HTM_F_OVERLAPS: REJECT_WHEN_AHEAD(h); if ((n = need(sp, C)) < 0) return n; return copy(d.overlap[qs], n, 4);
        case HTM_F_BOOSTED: REJECT_WHEN_AHEAD(h); if ((n = need(sp, C)) < 0) return n; return copy(d.boosted[qs], n, 8);
        case HTM_F_DUTY_CYCLE: REJECT_WHEN_AHEAD(h); if ((n = need(sp, C)) < 0) return n; return copy(d.duty, n, 4);
        case HTM_F_CELL_ACTIVATION: if ((n = need(tm, CW)) < 0) return n; return copy(d.act[q], n, 4);
        case HTM_F_CELL_PREDICTION: if ((n = need(tm, CW)) < 0) return n; return copy(d.pred[q], n, 4);
        case HTM_F_WINNER_WORDS: if ((n = need(tm, CW)) < 0) return n; return copy(d.win[q], n, 4);
        case HTM_F_BURSTING: if ((n = need(tm, d.k)) < 0) return n; return copy(d.bursting, n, 1);
        case HTM_F_SEG_NSYN: if ((n = need(tm, S)) < 0) return n; return copy(d.seg_nsyn, n, 4);
        case HTM_F_SEG_POTENTIAL:
        case HTM_F_MATCH_SEGMENT:
        case HTM_F_MATCH_INFO:
        case HTM_F_MATCH_JITTER: {
            // the device keeps one info word per segment; the matching-segment lists of
            // PredictiveProjection.State (ascending ids, projections.py:247) are its non-zero part
            if (!tm) { h->err = "htm_read: field not available on this handle"; return HTM_ERR_STATE; }
            if (field == HTM_F_SEG_POTENTIAL) {
                // the device keeps the potential of matching segments only: recompute all of them from the
                // last step's activation (same definition, projections.py:175-178 / :246)
                if ((n = need(true, S)) < 0) return n;
                if (S == 0) return 0;
                if (!c.has_distal) { memset(dst, 0, (size_t)S * 4); return S; }
                int *tmp = nullptr;
                if (hipMalloc((void **)&tmp, (size_t)S * 4) != hipSuccess) { h->err = "htm_read: hipMalloc failed"; return HTM_ERR_HIP; }
                hipLaunchKernelGGL(k_tm_potentials, dim3((unsigned)std::min<int64_t>((S * 8 + 255) / 256, 8192)), dim3(256), 0, h->stream, d, q, tmp, 0, (int)S);
                const bool ok = hipStreamSynchronize(h->stream) == hipSuccess && hipMemcpy(dst, tmp, (size_t)S * 4, hipMemcpyDeviceToHost) == hipSuccess;
                hipFree(tmp);
                if (!ok) { h->err = "htm_read: potentials kernel failed"; return HTM_ERR_HIP; }
                return S;
            }
            // one bit per segment says whether it is matching; info word and jitter exist for those only.  The
            // matching-segment lists of PredictiveProjection.State (ascending ids, projections.py:247):
            std::vector<uint32_t> bits(((size_t)S + 31) / 32, 0u);
            std::vector<uint32_t> info((size_t)S);
            std::vector<float> jit((size_t)S);
            if (S && c.has_distal && (hipMemcpy(bits.data(), d.match_bits[q], bits.size() * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                                      hipMemcpy(info.data(), d.seg_info, (size_t)S * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                                      hipMemcpy(jit.data(), d.seg_jit, (size_t)S * 4, hipMemcpyDeviceToHost) != hipSuccess)) { h->err = "htm_read: hipMemcpy failed"; return HTM_ERR_HIP; }
            auto is_matching = [&](int64_t i) { return (bits[(size_t)i >> 5] >> (i & 31)) & 1u; };
            int64_t m = 0;
            for (int64_t i = 0; i < S; ++i) m += is_matching(i);
            if ((n = need(true, m)) < 0) return n;
            m = 0;
            for (int64_t i = 0; i < S; ++i) {
                if (!is_matching(i)) continue;
                const uint32_t v = info[(size_t)i];
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
                    if (field == HTM_F_SEG_PRESYN) { int *v = (int *)dst + s * E + e; *v = valid ? enc_flat(*v & SYN_CELL, (int)K) : -1; }   // (without the connected flag)
                    else if (!valid) ((float *)dst)[s * E + e] = -1.0f;
                }
            return n;
        }
        case HTM_F_SEGCOUNT:
        case HTM_F_CELL_MAX_JITTER: {
            if ((n = need(tm, C * K)) < 0) return n;
            std::vector<uint32_t> tmp((size_t)C * KP);
            const void *src = field == HTM_F_SEGCOUNT ? (const void *)d.segcount : (const void *)d.cellmax[q];
            if (hipMemcpy(tmp.data(), src, tmp.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { h->err = "htm_read: hipMemcpy failed"; return HTM_ERR_HIP; }
            uint32_t *v = (uint32_t *)dst;
            for (int64_t col = 0; col < C; ++col)
                for (int64_t j = 0; j < K; ++j) v[col * K + j] = tmp[(size_t)col * KP + j];
            return n;
        }
        case HTM_F_SEG_GID: {
            if ((n = need(tm, S)) < 0) return n;
            if (d.seg_gid) return copy(d.seg_gid, n, 4);
            for (int64_t i = 0; i < n; ++i) ((int *)dst)[i] = (int)i;
            return n;
        }
        default: h->err = "htm_read: unknown field"; return HTM_ERR_ARGUMENT;
    }
}

// Rows [row_begin, row_begin + row_count) of a per-segment field: what htm_read returns for the whole pool, for a pool
// too large to read whole (configs[4]: 134 M rows).  HTM_F_MATCH_INFO comes back dense: one word per row, 0 = the row is
// not matching.
extern "C" int64_t htm_read_rows(htm_handle *h, int32_t field, int64_t row_begin, int64_t row_count, void *dst, int64_t count) {
    if (!h || !dst || row_begin < 0 || row_count < 0) return HTM_ERR_ARGUMENT;
    if (h) flush_tail(h);
    if (!h->cfg.enable_tm) { h->err = "htm_read_rows: handle has no Temporal Memory"; return HTM_ERR_STATE; }
    Counters c;
    int rc = read_counters(h, &c);
    if (rc) return rc;
    Dev &d = h->d;
    const int q = (int)((h->step_host + 1) & 1);
    const int64_t S = h->world > 1 ? c.L : c.S, E = d.E, K = d.K;
    if (row_begin + row_count > S) { h->err = "htm_read_rows: rows out of range"; return HTM_ERR_ARGUMENT; }
    const int64_t per = (field == HTM_F_SEG_PRESYN || field == HTM_F_SEG_PERM) ? E : 1, n = row_count * per;
    if (count < n) { h->err = "htm_read_rows: buffer too small"; return HTM_ERR_ARGUMENT; }
    if (n == 0) return 0;
    auto copy = [&](const void *src, size_t bytes) -> bool {
        if (hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) { h->err = "htm_read_rows: hipMemcpy failed"; return false; }
        return true;
    };
    switch (field) {
        case HTM_F_SEG_NSYN: return copy(d.seg_nsyn + row_begin, (size_t)n * 4) ? n : HTM_ERR_HIP;
        case HTM_F_SEG_GID: {
            if (d.seg_gid) return copy(d.seg_gid + row_begin, (size_t)n * 4) ? n : HTM_ERR_HIP;
            for (int64_t i = 0; i < n; ++i) ((int *)dst)[i] = (int)(row_begin + i);
            return n;
        }
        case HTM_F_SEG_CELL: {
            if (!copy(d.seg_cell + row_begin, (size_t)n * 4)) return HTM_ERR_HIP;
            for (int64_t i = 0; i < n; ++i) ((int *)dst)[i] = enc_flat(((int *)dst)[i], (int)K);
            return n;
        }
        case HTM_F_SEG_PRESYN:
        case HTM_F_SEG_PERM: {
            std::vector<int> nsyn((size_t)row_count);
            if (hipMemcpy(nsyn.data(), d.seg_nsyn + row_begin, (size_t)row_count * 4, hipMemcpyDeviceToHost) != hipSuccess) { h->err = "htm_read_rows: hipMemcpy failed"; return HTM_ERR_HIP; }
            if (!copy(field == HTM_F_SEG_PRESYN ? (const void *)(d.presyn + row_begin * E) : (const void *)(d.sperm + row_begin * E), (size_t)n * 4)) return HTM_ERR_HIP;
            for (int64_t s = 0; s < row_count; ++s)
                for (int64_t e = 0; e < E; ++e) {
                    const bool valid = e < (nsyn[(size_t)s] & ~(int)SEG_BUSY);
                    if (field == HTM_F_SEG_PRESYN) { int *v = (int *)dst + s * E + e; *v = valid ? enc_flat(*v & SYN_CELL, (int)K) : -1; }
                    else if (!valid) ((float *)dst)[s * E + e] = -1.0f;
                }
            return n;
        }
        case HTM_F_SEG_POTENTIAL: {
            if (!c.has_distal) { memset(dst, 0, (size_t)n * 4); return n; }
            int *tmp = nullptr;
            if (hipMalloc((void **)&tmp, (size_t)n * 4) != hipSuccess) { h->err = "htm_read_rows: hipMalloc failed"; return HTM_ERR_HIP; }
            hipLaunchKernelGGL(k_tm_potentials, dim3((unsigned)std::min<int64_t>((n * 8 + 255) / 256, 8192)), dim3(256), 0, h->stream, d, q, tmp,
                               (int)row_begin, (int)(row_begin + row_count));
            const bool ok = hipStreamSynchronize(h->stream) == hipSuccess && hipMemcpy(dst, tmp, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess;
            hipFree(tmp);
            if (!ok) { h->err = "htm_read_rows: potentials kernel failed"; return HTM_ERR_HIP; }
            return n;
        }
        case HTM_F_MATCH_INFO: {
            const int64_t w0 = row_begin >> 5, w1 = (row_begin + row_count + 31) >> 5;
            std::vector<uint32_t> bits((size_t)(w1 - w0), 0u);
            if (c.has_distal && hipMemcpy(bits.data(), d.match_bits[q] + w0, bits.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { h->err = "htm_read_rows: hipMemcpy failed"; return HTM_ERR_HIP; }
            if (!copy(d.seg_info + row_begin, (size_t)n * 4)) return HTM_ERR_HIP;
            for (int64_t i = 0; i < n; ++i) {
                const int64_t r = row_begin + i;
                const bool m = (bits[(size_t)((r >> 5) - w0)] >> (r & 31)) & 1u;
                ((uint32_t *)dst)[i] = m ? (((uint32_t *)dst)[i] & ~0x40000000u) : 0u;
            }
            return n;
        }
        default: h->err = "htm_read_rows: not a per-segment field"; return HTM_ERR_ARGUMENT;
    }
}

extern "C" int htm_write(htm_handle *h, int32_t field, const void *src, int64_t count) {
    if (!h || (!src && count > 0) || count < 0) return HTM_ERR_ARGUMENT;
    if (h) flush_tail(h);
    REJECT_WHEN_AHEAD(h);
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
    if (h->world > 1 && (field == HTM_F_SEG_CELL || field == HTM_F_SEG_NSYN || field == HTM_F_SEG_PRESYN || field == HTM_F_SEG_PERM)) {
        // a column-sharded handle is given the arrays of ALL segment ids (what an unsharded handle exports, or the merged
        // export of a sharded group) and keeps its own cells' rows at the commit
        const int64_t per = (field == HTM_F_SEG_PRESYN || field == HTM_F_SEG_PERM) ? E : 1;
        if (count > (int64_t)d.Scap * per) { h->err = "htm_write: too many elements"; return HTM_ERR_ARGUMENT; }
        if (field == HTM_F_SEG_PERM) { h->imp_perm.assign((const float *)src, (const float *)src + count); return HTM_OK; }
        std::vector<int> &v = field == HTM_F_SEG_CELL ? h->imp_seg_cell : field == HTM_F_SEG_NSYN ? h->imp_seg_nsyn : h->imp_presyn;
        v.assign((const int *)src, (const int *)src + count);
        if (field != HTM_F_SEG_NSYN) for (auto &x : v) x = x < 0 ? 0 : flat_enc(x, (int)K);
        return HTM_OK;
    }
    switch (field) {
        case HTM_F_DUTY_CYCLE: return put(d.duty, src, count, 4, C);
        case HTM_F_CELL_ACTIVATION: return put(d.act[q], src, count, 4, C * d.WPC);
        case HTM_F_CELL_PREDICTION: return put(d.pred[q], src, count, 4, C * d.WPC);
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
            if (field == HTM_F_WINNER_CELL) return put(d.winners[q], v.data(), count, 4, (int64_t)d.k * d.KP);
            if (field == HTM_F_SEG_CELL) return put(d.seg_cell, v.data(), count, 4, d.Scap);
            return put(d.presyn, v.data(), count, 4, (int64_t)d.Scap * E);
        }
        case HTM_F_SEGCOUNT:
        case HTM_F_CELL_MAX_JITTER: {
            if (count != C * K) { h->err = "htm_write: need column_dim * cell_dim elements"; return HTM_ERR_ARGUMENT; }
            std::vector<uint32_t> tmp((size_t)C * d.KP, 0u);
            const uint32_t *v = (const uint32_t *)src;
            for (int64_t col = 0; col < C; ++col)
                for (int64_t j = 0; j < K; ++j) tmp[(size_t)col * d.KP + j] = v[col * K + j];
            return put(field == HTM_F_SEGCOUNT ? (void *)d.segcount : (void *)d.cellmax[q], tmp.data(), (int64_t)tmp.size(), 4, (int64_t)tmp.size());
        }
        default: h->err = "htm_write: field is not writable"; return HTM_ERR_ARGUMENT;
    }
}

extern "C" int htm_import_begin(htm_handle *h, int64_t step_index) {
    if (!h || (step_index < 0 && step_index != HTM_IMPORT_PREV_STATE)) return HTM_ERR_ARGUMENT;
    if (h) flush_tail(h);
    REJECT_WHEN_AHEAD(h);
    if (h->world > 1 && step_index == HTM_IMPORT_PREV_STATE) { h->err = "prev_state adoption is not available on a column-sharded handle"; return HTM_ERR_STATE; }
    if (h->shard_open) { h->err = "htm_import_begin: a step opened with htm_shard_begin is not finished"; return HTM_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->import_keep = step_index == HTM_IMPORT_PREV_STATE;
    if (!h->import_keep) h->step_host = step_index;
    int rc = close_open_phases(h);
    if (rc) return rc;
    return HTM_OK;
}

// winner words (d.win[q], what HTM_F_WINNER_WORDS and State.winner_cell read) from the imported winner list
__global__ __launch_bounds__(256) void k_tm_winner_words(Dev d, int q, int n) {
    for (int c = blockIdx.x * 256 + threadIdx.x; c < d.C * d.WPC; c += gridDim.x * 256) d.win[q][c] = 0;
}
__global__ __launch_bounds__(256) void k_tm_winner_bits(Dev d, int q, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const int e = d.winners[q][i]; atomicOr(&d.win[q][e >> 5], 1u << (e & 31)); }
}

extern "C" int htm_import_commit(htm_handle *h, int32_t segments, int32_t matching_segments, int32_t winner_cells,
                                 int32_t has_distal_state, int32_t has_winner_cells) {
    if (!h) return HTM_ERR_ARGUMENT;
    if (h) flush_tail(h);
    Dev &d = h->d;
    const bool keep = h->import_keep;              // TemporalMemory.process(prev_state=X): the previous step's State only
    h->import_keep = false;
    Counters c;
    int rc = read_counters(h, &c);
    if (rc) return rc;
    if (keep) segments = c.S;                       // (the store stays what it is)
    if (segments < 0 || segments > d.Scap || matching_segments < 0 || matching_segments > segments ||
        winner_cells < 0 || winner_cells > d.k * d.KP) { h->err = "htm_import_commit: bad scalars"; return HTM_ERR_ARGUMENT; }
    const int q = (int)((h->step_host + 1) & 1);
    std::vector<int> g2l;                           // column-sharded: local row of every id this rank owns, -1 otherwise
    if (h->world > 1) {
        const size_t S = (size_t)segments, E = (size_t)d.E;
        if (h->imp_seg_cell.size() != S || h->imp_seg_nsyn.size() != S || h->imp_presyn.size() != S * E || h->imp_perm.size() != S * E) {
            h->err = "htm_import_commit: a column-sharded handle needs SEG_CELL / SEG_NSYN / SEG_PRESYN / SEG_PERM of all segment ids";
            return HTM_ERR_ARGUMENT;
        }
        g2l.assign((size_t)d.Scap, -1);
        std::vector<int> gid_of_row;
        std::vector<uint32_t> dead(((size_t)d.Scap + 31) / 32 + 32, 0u);
        const size_t nb1 = ((size_t)d.Scap + 1023) / 1024, nb2 = (nb1 + 1023) / 1024 + 1;
        std::vector<int> cnt1(nb1, 0), cnt2(nb2, 0);
        for (size_t g = 0; g < S; ++g) {
            if (h->imp_seg_nsyn[g] < d.match_thr) { dead[g >> 5] |= 1u << (g & 31); cnt1[g >> 10] += 1; cnt2[g >> 20] += 1; }
            const int col = h->imp_seg_cell[g] >> 5;
            if (col >= d.c0 && col < d.c1) { g2l[g] = (int)gid_of_row.size(); gid_of_row.push_back((int)g); }
        }
        const size_t L = gid_of_row.size();
        if (L > (size_t)d.Lcap) { h->err = "htm_import_commit: more segments of this rank's cells than segment_capacity_local"; return HTM_ERR_CAPACITY; }
        std::vector<int> cell(L), nsyn(L), presyn(L * E);
        std::vector<float> perm(L * E);
        for (size_t r = 0; r < L; ++r) {
            const size_t g = (size_t)gid_of_row[r];
            cell[r] = h->imp_seg_cell[g];
            nsyn[r] = h->imp_seg_nsyn[g];
            memcpy(&presyn[r * E], &h->imp_presyn[g * E], E * 4);
            memcpy(&perm[r * E], &h->imp_perm[g * E], E * 4);
        }
        HIPCHK(h, hipMemset(d.seg_gid, 0xFF, (size_t)d.Lcap * 4));
        HIPCHK(h, hipMemset(d.seg_nsyn, 0, (size_t)d.Lcap * 4));
        if (L) {
            HIPCHK(h, hipMemcpy(d.seg_gid, gid_of_row.data(), L * 4, hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(d.seg_cell, cell.data(), L * 4, hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(d.seg_nsyn, nsyn.data(), L * 4, hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(d.presyn, presyn.data(), L * E * 4, hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(d.sperm, perm.data(), L * E * 4, hipMemcpyHostToDevice));
        }
        HIPCHK(h, hipMemcpy(d.g2l, g2l.data(), g2l.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(d.dead_bits, dead.data(), dead.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(d.recyc_cnt, cnt1.data(), cnt1.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(d.recyc_cnt2, cnt2.data(), cnt2.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemset(d.dead_list, 0, sizeof(int)));
        c.L = (int32_t)L;
        c.n_lfree = 0;
        h->imp_seg_cell.clear(); h->imp_seg_nsyn.clear(); h->imp_presyn.clear(); h->imp_perm.clear();
        h->imp_seg_cell.shrink_to_fit(); h->imp_presyn.shrink_to_fit(); h->imp_perm.shrink_to_fit();
    }
    const int rows = h->world > 1 ? c.L : segments;     // rows of the per-row arrays (info words, jitter, match bits)
    auto row_of = [&](int gid) { return h->world > 1 ? g2l[(size_t)gid] : gid; };
    if (!keep) {
        c.step[h->step_host & 1] = (uint32_t)h->step_host;
        c.n_work[0] = c.n_work[1] = 0;
        c.n_bind[0] = c.n_bind[1] = 0;
        c.S = segments;
        h->seg_hint = rows;                         // (the one place where the count can go down)
        if (h->seg_pinned) *h->seg_pinned = rows;
    }
    {   // dense per-segment info from the staged PredictiveProjection.State lists
        const size_t M = (size_t)matching_segments;
        if (has_distal_state && (h->imp_pot.size() != (size_t)segments || h->imp_match_seg.size() != M ||
                                 h->imp_match_info.size() != M || h->imp_match_jit.size() != M)) {
            h->err = "htm_import_commit: SEG_POTENTIAL / MATCH_* fields missing or of the wrong length";
            return HTM_ERR_ARGUMENT;
        }
        std::vector<uint32_t> info((size_t)rows, 0u), bits((size_t)(d.Lcap + 255) / 256 * 8, 0u);
        std::vector<float> jit((size_t)rows, 0.f);
        if (has_distal_state) {
            for (size_t i = 0; i < M; ++i) {
                if (h->imp_match_seg[i] < 0 || h->imp_match_seg[i] >= segments) { h->err = "htm_import_commit: matching segment id out of range"; return HTM_ERR_ARGUMENT; }
                const int sgm = row_of(h->imp_match_seg[i]);
                if (sgm < 0) continue;              // (another rank's segment)
                info[(size_t)sgm] = (h->imp_match_info[i] & ~0x40000000u) | 0x40000000u;
                jit[(size_t)sgm] = h->imp_match_jit[i];
                bits[(size_t)sgm >> 5] |= 1u << (sgm & 31);
            }
        }
        if (rows) {
            HIPCHK(h, hipMemcpy(d.seg_info, info.data(), info.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(d.seg_jit, jit.data(), jit.size() * 4, hipMemcpyHostToDevice));
        }
        // all of the bitmap: the import is the one place where the segment count can shrink (rollback to an
        // earlier checkpoint), and ids at or above it must read "not matching", as the classification assumes
        HIPCHK(h, hipMemcpy(d.match_bits[q], bits.data(), bits.size() * 4, hipMemcpyHostToDevice));
        // the other buffers (what the coming step's scan accumulates into) start clean
        HIPCHK(h, hipMemset(d.match_bits[q ^ 1], 0, bits.size() * 4));
        HIPCHK(h, hipMemset(d.cellmax[q ^ 1], 0, (size_t)d.C * d.KP * 4));
        h->imp_pot.clear(); h->imp_match_seg.clear(); h->imp_match_info.clear(); h->imp_match_jit.clear();
    }
    c.n_win[q] = winner_cells;
    c.has_winner[q] = has_winner_cells ? 1 : 0;
    c.has_distal = has_distal_state ? 1 : 0;
    c.cm_dense_step = (uint32_t)h->step_host + 1u;
    if (!keep) h->window_known = false;
    if (!keep) c.error = 0;                         // a checkpoint restore starts clean; an adopted previous State does not
                                                    // forgive an overflow of the store it keeps
    HIPCHK(h, hipMemcpy(d.ctr, &c, sizeof(c), hipMemcpyHostToDevice));
    if (h->cfg.enable_tm) {
        // the previous step's active columns (State.active_cell / winner_cell index with them): the columns with an active cell
        {
            std::vector<uint32_t> act((size_t)d.C * d.WPC);
            HIPCHK(h, hipMemcpy(act.data(), d.act[q], act.size() * 4, hipMemcpyDeviceToHost));
            std::vector<int> cols((size_t)d.k, 0);
            int n = 0;
            for (int col = 0; col < d.C && n < d.k; ++col)
                if (act[(size_t)col * d.WPC] | act[(size_t)col * d.WPC + d.WPC - 1]) cols[(size_t)n++] = col;
            HIPCHK(h, hipMemcpy(d.active_cols[q], cols.data(), cols.size() * 4, hipMemcpyHostToDevice));
        }
        // State.winner_cell / HTM_F_WINNER_WORDS read the winner words: rebuild them from the imported list
        hipLaunchKernelGGL(k_tm_winner_words, dim3(std::min((d.C + 255) / 256, 1024)), dim3(256), 0, h->stream, d, q, winner_cells);
        if (winner_cells > 0) hipLaunchKernelGGL(k_tm_winner_bits, dim3((winner_cells + 255) / 256), dim3(256), 0, h->stream, d, q, winner_cells);
        if (!keep) {
            if (h->world == 1) {                    // (a shard's counts came with its dead bits, above)
                HIPCHK(h, hipMemsetAsync(d.recyc_cnt2, 0, ((size_t)(h->s1024_blocks + 1023) / 1024 + 1) * sizeof(int), h->stream));
                hipLaunchKernelGGL(k_tm_recount, dim3(h->s1024_blocks), dim3(256), 0, h->stream, d);
            }
            hipLaunchKernelGGL(k_tm_flag_connected, dim3(std::min(4096, std::max(1, (int)(((long long)rows * d.E + 255) / 256)))), dim3(256), 0, h->stream, d);
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return HTM_OK;
}

extern "C" int htm_profile(htm_handle *h, int32_t enable) {
    if (!h) return HTM_ERR_ARGUMENT;
    if (h) flush_tail(h);
    h->profile = enable != 0;
    return HTM_OK;
}

extern "C" int64_t htm_trace_read(htm_handle *h, uint64_t *dst, int64_t count) {
    if (!h || !dst) return HTM_ERR_ARGUMENT;
    if (h) flush_tail(h);
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
    if (h) flush_tail(h);
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
