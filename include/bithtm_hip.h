/*
 * bithtm_hip.h -- C ABI of the MI355X-native bitHTM timestep engine (libbithtm_hip.so).
 *
 * The reference (cokwa/bitHTM) is pure Python/NumPy and has no FFI of its own; its drop-in
 * boundary is the Python class surface bithtm/networks.py:7-149.  This header is the
 * boundary a binding for that surface uses: one handle owns all device state of one
 * SpatialPooler + TemporalMemory pair, one call enqueues one timestep, results are read
 * back lazily.  Plain pointers and sizes only; no C++ or torch types cross it.
 *
 * Every entry point names the reference interface it replaces (file:line relative to the
 * reference checkout).  All functions return 0 on success or a negative htm_status; after
 * a failure htm_last_error() describes it.  A handle is not thread-safe; different
 * handles are independent.  Host buffers are borrowed for the duration of the call only.
 *
 * Cell ids crossing this ABI are the reference's flat ids  col * cell_dim + cell
 * (networks.py:67-71).  Bitmaps are one uint32 word per column, bit j = cell j.
 */
#ifndef BITHTM_HIP_H
#define BITHTM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BITHTM_ABI_VERSION 4

typedef struct htm_handle htm_handle;

typedef enum htm_status {
    HTM_OK = 0,
    HTM_ERR_ARGUMENT = -1,      /* bad pointer / size / field id / config */
    HTM_ERR_HIP = -2,           /* a HIP runtime call failed */
    HTM_ERR_CAPACITY = -3,      /* segment pool or synapse slots exhausted (reported, never silent) */
    HTM_ERR_STATE = -4          /* call not valid in the handle's current state */
} htm_status;

/* Scalars the reference forms implicitly, with the dtype each one ends up in
 * (projections.py:7-11,24,102,205-210; regularizations.py:5-9,16,20-21).  The Python binding
 * computes them with the same Python expressions so they are bit-identical. */
typedef struct htm_config {
    uint32_t struct_bytes;              /* sizeof(htm_config), checked */
    int32_t device;                     /* HIP device ordinal */
    int32_t input_dim;                  /* SpatialPooler.input_dim   (networks.py:18) */
    int32_t column_dim;                 /* column_dim                (networks.py:19,52) */
    int32_t cell_dim;                   /* TemporalMemory.cell_dim   (networks.py:53), 1..64; a column-sharded handle: 1..32 (a model with
                                           still more cells per column keeps its segment store on an engine of ceil(N / 32) columns of
                                           32 cells: DESIGN.md section 7) */
    int32_t active_columns;             /* k                         (networks.py:20,137) */
    int32_t enable_sp;                  /* 0: handle is a stand-alone TemporalMemory */
    int32_t enable_tm;                  /* 0: handle is a stand-alone SpatialPooler */
    /* DenseProjection (projections.py:6-24) */
    double sp_permanence_threshold;     /* permanence >= threshold  <=> connected (:19) */
    double sp_delta_on;                 /* 1.0*(inc+dec)-dec         (:24) */
    double sp_delta_off;                /* 0.0*(inc+dec)-dec         (:24) */
    /* ExponentialBoosting (regularizations.py:4-21) */
    float boost_coefficient;            /* float32(-(intensity/density))  (:16) */
    float duty_momentum;                /* float32(momentum)              (:20) */
    float duty_increment;               /* float32(1.0 - momentum)        (:21) */
    /* PredictiveProjection / SparseProjection (projections.py:97-109,205-293) */
    double tm_learn_active;             /* 1.0*(a-b)+b with a=+increment, b=-decrement (:102,:287) */
    double tm_learn_inactive;           /* 0.0*(a-b)+b */
    double tm_punish_active;            /* same with a=-punishment, b=0.0             (:292) */
    double tm_punish_inactive;
    int32_t tm_learn_prune;             /* min(a,b) < 0 (:105) */
    int32_t tm_punish_prune;
    float tm_permanence_initial;        /* float32(permanence_initial)   (:158) */
    float tm_permanence_threshold;      /* float32(permanence_threshold) (:171) */
    int32_t segment_activation_threshold;   /* (:221) */
    int32_t segment_matching_threshold;     /* (:222) */
    int32_t segment_sampling_synapses;      /* (:223), 1..64 */
    /* fixed-capacity pool replacing DynamicArray2D growth (utils.py:79-135) */
    int32_t segment_capacity;           /* max segment ids (of the whole model); overflow => HTM_ERR_CAPACITY */
    int32_t segment_capacity_local;     /* column-sharded handles: rows for the segments of this rank's own cells
                                           (0 = 2 * segment_capacity / shard_world, at most segment_capacity) */
    int32_t segment_slots;              /* synapse slots per segment, multiple of 64, <= 512 */
    uint32_t seed;                      /* keyed random draws, see bithtm_amd/csrc/htm_rng.h */
    int32_t shard_rank;                 /* column sharding: this handle owns columns               */
    int32_t shard_world;                /*   [rank, rank+1) * column_dim / world; 0 or 1 = unsharded */
    int32_t use_caller_stream;          /* 1: enqueue on `stream` (NULL then means the default stream, e.g.
                                           torch's current stream); 0: create a private stream */
    void *stream;                       /* hipStream_t, see use_caller_stream */
} htm_config;

typedef struct htm_info {
    int64_t step_index;                 /* timesteps processed */
    int32_t segments;                   /* S: allocated segment ids (len(segment_bundle)) */
    int32_t local_segments;             /* rows in use on this handle: = segments, except on a column-sharded handle
                                           (rows of the segments its own cells own, some of them free) */
    int32_t matching_segments;          /* len(distal_state.matching_segment) of the last step */
    int32_t winner_cells;               /* len(winner_cell[0]) of the last step */
    int32_t active_cells;               /* len(active_cell[0]) of the last step */
    int32_t has_distal_state;           /* last_state.distal_state is not None */
    int32_t has_winner_cells;           /* last_state.winner_cell is not None */
    int32_t capacity_error;             /* sticky: 1 = segment pool, 2 = synapse slots, 4 = work list / growth staging,
                                           8 = dead-segment report of a sharded handle, 16 = (internal) a block of the
                                           in-kernel select exchange never arrived: the step's result is invalid */
    int32_t words_per_row;              /* packed input words per SP row (input_dim padded to 128 bits) */
    int32_t new_segment_requests;       /* last step: winners without a matching segment (projections.py:271) */
    int32_t recycled_segments;          /* last step: of those, served by recycling (projections.py:80-85) */
    int32_t appended_segments;          /* last step: served by fresh ids (projections.py:90-94) */
    int32_t work_items;                 /* last step: segments that learned or were punished */
    int32_t select_fallbacks;           /* steps so far whose top-k select overflowed the per-block records
                                           and took the exact in-kernel fallback (slower, same result) */
    int32_t candidate_exact_steps;      /* column-sharded handles: steps so far whose LOCAL select cut its threshold bin exactly
                                           instead of handing the whole bin over (slower, same result) */
    int32_t hot_select_steps;           /* column-sharded handles: steps so far whose GLOBAL select was settled among the ranks'
                                           hot lists (the candidates near the previous step's k-th key) without reading the
                                           other candidates */
    int32_t select_zoom_steps;          /* steps so far whose top-k select found its threshold bin crowded (blocks holding several
                                           distinct keys each) and cut it to the k-th key's sub-bin before ranking it */
} htm_info;

/* Device arrays readable with htm_read / writable with htm_write. Element type and count
 * (C = column_dim, k = active_columns, N = C * cell_dim, S = htm_info.segments,
 * E = segment_slots, M = htm_info.matching_segments, Wn = htm_info.winner_cells, W = 32-bit words of cells per column:
 * 1 for cell_dim <= 32, 2 up to 64 -- cell j of column c is bit j % 32 of word c * W + j / 32). */
typedef enum htm_field {
    HTM_F_ACTIVE_COLUMN = 1,   /* int32[k]   State.active_column, ascending (networks.py:29) */
    HTM_F_OVERLAPS = 2,        /* int32[C]   State.overlaps (networks.py:27) */
    HTM_F_BOOSTED = 3,         /* double[C]  State.boosted_overlaps (networks.py:28) */
    HTM_F_DUTY_CYCLE = 4,      /* float[C]   ExponentialBoosting.duty_cycle (regularizations.py:13) */
    HTM_F_CELL_ACTIVATION = 5, /* uint32[C*W] State.cell_activation, packed (networks.py:118-119) */
    HTM_F_CELL_PREDICTION = 6, /* uint32[C*W] State.cell_prediction, packed (networks.py:122) */
    HTM_F_WINNER_WORDS = 7,    /* uint32[C*W] winner cells, packed (networks.py:102) */
    HTM_F_BURSTING = 8,        /* uint8[k]   State.active_column_bursting (networks.py:97) */
    HTM_F_WINNER_CELL = 9,     /* int32[Wn]  flat winner cell ids, ascending (networks.py:103-104) */
    HTM_F_SEG_CELL = 10,       /* int32[S]   segment_bundle (projections.py:226) */
    HTM_F_SEG_NSYN = 11,       /* int32[S]   output_edges (projections.py:42) */
    HTM_F_SEG_PRESYN = 12,     /* int32[S*E] presynaptic flat cell id per slot, -1 = free */
    HTM_F_SEG_PERM = 13,       /* float[S*E] output_permanence (projections.py:44), -1.0 = free */
    HTM_F_SEGCOUNT = 14,       /* int32[N]   bundle_segments (projections.py:227) */
    HTM_F_SEG_POTENTIAL = 15,  /* int32[S]   State.segment_potential (projections.py:246) */
    HTM_F_MATCH_SEGMENT = 16,  /* int32[M]   State.matching_segment, UNORDERED (projections.py:247) */
    HTM_F_MATCH_INFO = 17,     /* uint32[M]  potential | activation<<12 | active<<31, same order */
    HTM_F_MATCH_JITTER = 18,   /* float[M]   matching_segment_jittered_potential, same order */
    HTM_F_CELL_MAX_JITTER = 19,/* float[N]   State.max_jittered_potential (projections.py:236-238) */
    HTM_F_SEG_GID = 20         /* int32[S]   global segment id of each row: 0..S-1, except on a column-sharded handle,
                                             where the per-segment fields above have htm_info.local_segments rows
                                             (the segments of the rank's own cells; -1 = free row) */
} htm_field;

/* Construction: HierarchicalTemporalMemory.__init__ / SpatialPooler.__init__ /
 * TemporalMemory.__init__ (networks.py:14-24,48-57,132-144).  SP permanences start at zero;
 * upload the matrix drawn as in projections.py:16 with htm_sp_set_permanence. */
int htm_create(const htm_config *config, htm_handle **out);
void htm_destroy(htm_handle *h);
const char *htm_last_error(const htm_handle *h);     /* h may be NULL: error of the last failed htm_create */
int htm_abi_version(void);

/* DenseProjection.permanence (projections.py:16): rows [row_begin, row_begin+row_count) of the
 * float64 [column_dim, input_dim] matrix, row-major, no padding. set rebuilds the connected mask. */
int htm_sp_set_permanence(htm_handle *h, const double *rows, int32_t row_begin, int32_t row_count);

/* TemporalMemory.process(..., epsilon=1e-8) (networks.py:91): the tolerance, compared as float32 the way NumPy compares
 * a Python scalar with float32 arrays, of the "best matching" and "least used" ties (networks.py:81,88) and of the
 * "best matching segment" test of learning (projections.py:267).  0 < epsilon <= 1 (the reference's other uses --
 * prediction > epsilon, max potential < epsilon -- then mean what they mean at 1e-8); stays until set again. */
int htm_set_epsilon(htm_handle *h, float epsilon);
int htm_sp_get_permanence(htm_handle *h, double *rows, int32_t row_begin, int32_t row_count);

/* One timestep, enqueued asynchronously.  packed_input: input bit i is bit (i & 31) of word
 * (i >> 5); ceil(input_dim / 32) host words.
 *   htm_step     HierarchicalTemporalMemory.process(input, learning)   (networks.py:146-149)
 *   htm_sp_step  SpatialPooler.process(input, learning)                (networks.py:26-35)
 *   htm_tm_step  TemporalMemory.process(sp_state, learning=, return_winner_cell=)
 *                (networks.py:91-128) for a stand-alone TM: active_column is a host array of n
 *                distinct column ids (any order; processed in ascending order).
 * htm_step may hold the step's last launch (learning + segment scan) back so that it rides in one of the next htm_step's
 * launches (beside its select finish, or beside its overlap); every other entry point lets it go before doing anything else (htm_sync included), so the only way to observe
 * it is to synchronise the STREAM yourself between two htm_step calls -- call htm_sync instead. */
int htm_step(htm_handle *h, const uint32_t *packed_input, int32_t learning);
int htm_sp_step(htm_handle *h, const uint32_t *packed_input, int32_t learning);
int htm_tm_step(htm_handle *h, const int32_t *active_column, int32_t n, int32_t learning,
                int32_t return_winner_cell);

/* PredictiveProjection.update / .process (projections.py:257-293, :245-255) called on their own -- a caller that writes its
 * own TemporalMemory.process around the device's segment store (its own winner-cell rule, its own punishment mask).
 *   htm_tm_update  learning: columns[i] (distinct, at most active_columns of them) has the learning cells winner_words[i]
 *                  (bit j = cell j: `learning_output` / `output_learning`; cell_dim above 32: two words per listed column, side
 *                  by side, as in the htm_field arrays), of which unaccounted_words[i] get a new segment
 *                  (:271-281); punish_words = `output_punishment` as one word (two) per column of the model, or NULL = every cell
 *                  of a column not listed (what TemporalMemory.process passes, networks.py:107-108,111).  prev_state,
 *                  input_activation and winner_input of the reference's signature are the handle's previous step (its own
 *                  last one, or whatever was written with htm_import_begin(HTM_IMPORT_PREV_STATE) / htm_write).
 *   htm_tm_scan    the scan against the cells of active_words (one word -- two -- per column of the model), which become the
 *                  step's cell activation; closes the timestep.  PredictiveProjection.State is read with htm_read. */
int htm_tm_update(htm_handle *h, const int32_t *columns, const uint32_t *winner_words, const uint32_t *unaccounted_words,
                  int32_t n, const uint32_t *punish_words);
int htm_tm_scan(htm_handle *h, const uint32_t *active_words);

/* SpatialPooler.process (networks.py:26-35) one phase per call, for handles whose plug-in objects (proximal_projection=,
 * boosting=, inhibition=: networks.py:16,22-24) partly live on the host: the binding interleaves these calls with the
 * user's `process` / `update` methods.  The phases work on the current timestep and do not close it: on a handle with
 * a Temporal Memory htm_tm_step (with the winner list) does, on a Spatial Pooler alone HTM_SP_COMMIT.  Results are
 * read with htm_read (HTM_F_OVERLAPS / _BOOSTED / _ACTIVE_COLUMN).  Each phase is also the stand-alone form of the
 * reference method named beside it. */
typedef enum htm_sp_phase_id {
    HTM_SP_OVERLAP = 1, /* DenseProjection.process (projections.py:18-21) + ExponentialBoosting.process
                           (regularizations.py:15-17); data = packed input as for htm_step */
    HTM_SP_BOOST = 2,   /* ExponentialBoosting.process on overlaps computed elsewhere; data = int32[column_dim] */
    HTM_SP_SELECT = 3,  /* GlobalInhibition.process (regularizations.py:28-29) on the device's boosted overlaps
                           (data = NULL) or on boosted overlaps computed elsewhere (data = double[column_dim]) */
    HTM_SP_ACTIVE = 4,  /* a winner list chosen elsewhere; data = int32[count] distinct columns, count <= active_columns */
    HTM_SP_LEARN = 5,   /* DenseProjection.update (projections.py:23-24) on the current winner list; data = packed
                           input, or NULL: the input of HTM_SP_OVERLAP */
    HTM_SP_DUTY = 6,    /* ExponentialBoosting.update (regularizations.py:19-21) on the current winner list */
    HTM_SP_COMMIT = 7   /* close the timestep of a handle without Temporal Memory */
} htm_sp_phase_id;
int htm_sp_phase(htm_handle *h, int32_t phase, const void *data, int64_t count);

/* n_steps timesteps of htm_step over a bank of n_inputs packed inputs that is ALREADY IN
 * DEVICE MEMORY (words_per_row words each, see htm_info); step t reads input
 * (step_index % n_inputs).  Nothing is copied or synchronised: this is the loop
 * example.py:48-53 runs, with the input bank resident in HBM.  use_graph bit 0: replay captured
 * hipGraphs (one per step, or per 16 steady-state steps) instead of issuing the launches one by one;
 * bit 1: do NOT pipeline.  By default the Spatial Pooler works ahead of the Temporal Memory inside the
 * call -- the next step's overlaps and winner list are computed in the two launches of the current TM step
 * (its permanence and duty-cycle updates are not ahead; in the four-launch schedule a handle falls back to when the
 * scan's column bitmap does not fit the LDS they are) --; it never looks past n_steps, so the state a call leaves
 * behind is exactly that of n_steps htm_step calls. */
int htm_run(htm_handle *h, const uint32_t *device_inputs, int32_t n_inputs, int32_t n_steps,
            int32_t learning, int32_t use_graph);
/* use_graph bit 2 (HTM_RUN_CONTINUE): a caller that streams its input in chunks promises that the next call is another
 * htm_run on the same bank, n_inputs and learning flag.  The Spatial Pooler then keeps working ahead across the end
 * of this call (the next step's overlaps and winner list are computed beside this call's last Temporal Memory
 * step) and the next call starts in the steady state instead of with a cold start of three more launches.  Until a
 * later htm_run ends without the bit, every other call that needs the Spatial Pooler's state (htm_step, htm_sp_*,
 * htm_tm_step, state import, the Spatial Pooler fields of htm_read) returns HTM_ERR_STATE; the Temporal Memory's state
 * is that of exactly the steps run so far.  Where the pipelined schedule is not available the bit is ignored; a handle
 * that is ahead when the schedule becomes unavailable (another handle with its own stream appears on the device)
 * finishes the step it had begun in the schedule it began it in and goes on unpipelined -- never an error. */
#define HTM_RUN_GRAPH 1
#define HTM_RUN_NO_PIPELINE 2
#define HTM_RUN_CONTINUE 4

/* Capture and instantiate, without running anything, every hipGraph the htm_run call with the same
 * arguments will replay when it comes next (graphs are otherwise built lazily inside htm_run, the first
 * time a launch pattern is met).  Latency-sensitive callers invoke it once after their warm-up. */
int htm_prepare(htm_handle *h, const uint32_t *device_inputs, int32_t n_inputs, int32_t n_steps,
                int32_t learning, int32_t use_graph);

/* What an htm_run / htm_prepare call with these arguments would do if it came now (nothing is enqueued): a caller that
 * reports how it ran (bench.py's `config.hip_graph`) asks instead of assuming.  Bits of the result:
 *   HTM_PLAN_GRAPH      the steady-state steps replay hipGraphs (use_graph bit 0 AND n_steps at or above the eager limit,
 *                       64 unless BITHTM_EAGER_BELOW says otherwise, AND no htm_profile collection in progress)
 *   HTM_PLAN_PIPELINED  the Spatial Pooler works ahead of the Temporal Memory (heterogeneous launches)
 *   HTM_PLAN_LEAN       ... in the two-launch schedule (three under BITHTM_LEAN=1; else four launches per step)
 *   HTM_PLAN_SCAN_LARGE the segment scan runs in its streaming (large-pool) form
 * Negative: an error status. */
#define HTM_PLAN_GRAPH 1
#define HTM_PLAN_PIPELINED 2
#define HTM_PLAN_LEAN 4
#define HTM_PLAN_SCAN_LARGE 8
int htm_run_plan(htm_handle *h, int32_t n_steps, int32_t use_graph);

/* Convenience for callers without their own device allocator: copy n_inputs packed inputs
 * (ceil(input_dim/32) host words each, as for htm_step) into a handle-owned device bank laid out
 * as htm_run expects, and return its device address.  Freed by htm_destroy. */
int htm_bank_upload(htm_handle *h, const uint32_t *host_inputs, int32_t n_inputs, uint32_t **device_bank);

/* Column-sharded timestep (one handle per GPU, shard_world > 1; DESIGN.md "Multi-GPU").  The
 * reference has no counterpart: it is HierarchicalTemporalMemory.process (networks.py:146-149)
 * split around the one exchange the sharding needs.
 *   htm_shard_begin   the rank's own part that precedes the exchange; writes this rank's record
 *                     (htm_shard_record_bytes bytes) to send_device.  The input is either a
 *                     device bank as for htm_run (packed_input == NULL) or one host input as for
 *                     htm_step (device_inputs == NULL).
 *   (caller)          all-gather of the records in rank order into recv_device
 *                     (world * record bytes) on the same stream, e.g. RCCL ncclAllGather
 *   htm_shard_finish  global top-k, segment allocation and this rank's share of the learning and
 *                     of the segment scan.
 * With identical inputs on all ranks the union of the ranks' results equals the unsharded result
 * bit for bit. */
int64_t htm_shard_record_bytes(htm_handle *h);
int htm_shard_begin(htm_handle *h, const uint32_t *device_inputs, int32_t n_inputs, const uint32_t *packed_input,
                    int32_t learning, void *send_device);
int htm_shard_finish(htm_handle *h, const void *recv_device, int32_t learning);

/* The same timestep as ONE call, with the exchange done inside the library by RCCL (ncclAllGather on the
 * handle's stream, device to device over xGMI; librccl is loaded when first needed).  One process per GPU:
 *   rank 0:     htm_shard_unique_id(id)          128 bytes, handed to the other ranks by the caller
 *   every rank: htm_shard_comm_init(h, id)       collective (ncclCommInitRank); allocates the record buffers
 *   every rank: htm_shard_step(h, ...)           per timestep; the input as for htm_shard_begin */
int htm_shard_unique_id(void *out128);
int htm_shard_comm_init(htm_handle *h, const void *unique_id128);
int htm_shard_step(htm_handle *h, const uint32_t *device_inputs, int32_t n_inputs, const uint32_t *packed_input,
                   int32_t learning);
int htm_shard_comm_size(htm_handle *h);              /* ranks of that communicator (ncclCommCount), or a negative status */
int htm_shard_graph_ok(htm_handle *h);               /* 1: htm_shard_comm_init's preflight (a record-sized all-gather over this
                                                        communicator, checked, then captured into a hipGraph, replayed and checked
                                                        again) passed and htm_shard_run replays graphs; 0: it launches eagerly */
/* n_steps of htm_shard_step over a bank resident in device memory, without the host in the loop: the launches of whole
 * timesteps, the collective included, are replayed as hipGraphs (use_graph bit 0; RCCL's all-gather is captured like a
 * kernel -- where the runtime refuses, the call launches eagerly instead), and inside the call the overlap of step t + 1 on
 * the rank's own columns rides in the last launch of step t (bit 1, HTM_RUN_NO_PIPELINE: not).  The state a call leaves
 * behind is that of n_steps htm_shard_step calls.  htm_shard_group_run: the same for all the ranks of a group inside one
 * process (see htm_shard_group_step). */
int htm_shard_run(htm_handle *h, const uint32_t *device_inputs, int32_t n_inputs, int32_t n_steps, int32_t learning, int32_t use_graph);
int htm_shard_group_run(htm_handle *const *handles, int32_t n, const uint32_t *const *device_inputs, int32_t n_inputs, int32_t n_steps,
                        int32_t learning, int32_t use_graph);
/* RCCL round trip at world size 1 on `device` (communicator, all-gather on a stream, compare; then the same collective
 * captured into a hipGraph and replayed): what a box with one GPU can verify of the path htm_shard_step / htm_shard_run
 * take.  0: both work; 1: the collective works but is not capturable on this runtime; negative: failure. */
int htm_rccl_selftest(int32_t device);

/* The keyed random draws of the engine (bithtm_amd/csrc/htm_rng.h), computed on the host: out[i] =
 * the number in [0, 1) -- a multiple of 2^-24 -- that the step `step` of a handle created with `seed` uses for
 *   stream 1  the least-used-cell jitter of flat cell a[i]                                     (networks.py:87)
 *   stream 2  the growth priority of presynaptic cell b[i] for segment a[i]                    (projections.py:120)
 *   stream 3  the jitter of matching segment a[i]                                              (projections.py:235)
 * (b may be NULL: all zeros).  This is the direction in which the reference and the engine are made to agree draw for
 * draw: the reference consumes THESE numbers where it would call np.random.rand (INTEGRATION.md section 5 shows the patch);
 * the engine cannot take MT19937's instead -- their shapes depend on data the step has not produced yet when it starts. */
int htm_keyed_draws(uint32_t seed, int32_t stream, uint32_t step, const uint32_t *a, const uint32_t *b, int64_t n, double *out);

/* All the shards of one model inside ONE process on one device: handles[r] = rank r of n = shard_world, created on
 * the same stream; the all-gather becomes n x n device copies.  For tests and single-GPU rehearsals of the sharded
 * path.  device_inputs[r] = rank r's copy of the input bank (or NULL and one host input, as for htm_step). */
int htm_shard_group_step(htm_handle *const *handles, int32_t n, const uint32_t *const *device_inputs, int32_t n_inputs,
                         const uint32_t *packed_input, int32_t learning);

/* Pre-populated segment pool, generated on the device (BASELINE.json configs[4]: 255 segments per cell, a pure
 * scan stress): every cell with flat id in [cell_begin, cell_end) gets segments_per_cell segments of `synapses`
 * synapses (>= the matching threshold, <= 64) to keyed-random distinct presynaptic cells with permanences keyed-
 * uniform in [perm_lo, perm_hi); segment ids are (cell - cell_begin) * segments_per_cell + j.  Fresh handles only.
 * On a column-sharded handle only the rows of its own cells are generated; give every handle of the group the
 * same range.  (The reference grows its store step by step, projections.py:226; it has no counterpart.) */
int htm_populate(htm_handle *h, int64_t cell_begin, int64_t cell_end, int32_t segments_per_cell, int32_t synapses,
                 double perm_lo, double perm_hi, uint32_t seed);

int htm_sync(htm_handle *h);
/* the hipStream_t the handle enqueues on (its own, or the caller's: htm_config.use_caller_stream) -- for callers that order
 * their own work against it, and for creating further handles on the same stream */
int htm_get_stream(htm_handle *h, void **stream);
int htm_get_info(htm_handle *h, htm_info *out);      /* synchronises; HTM_ERR_CAPACITY (with *out filled
                                                         in) once a fixed-capacity pool has overflowed */

/* Lazy read-back of State fields / state export (synchronises); count = number of ELEMENTS
 * the caller's buffer holds; it must be >= the field's current element count. Returns the
 * number of elements written (>= 0) or a negative status. */
int64_t htm_read(htm_handle *h, int32_t field, void *dst, int64_t count);

/* Rows [row_begin, row_begin + row_count) of a per-segment field (HTM_F_SEG_CELL / _NSYN / _PRESYN / _PERM / _POTENTIAL /
 * _GID, same element types as htm_read), for pools too large to read whole (the reference's `segment_bundle[a:b]`,
 * `output_edge[a:b]`, `output_permanence[a:b]`: projections.py:226,43-44).  HTM_F_MATCH_INFO comes back DENSE here: one
 * word per row, potential | activation<<12 | active<<31 for a matching row, 0 otherwise.  count = elements dst holds. */
int64_t htm_read_rows(htm_handle *h, int32_t field, int64_t row_begin, int64_t row_count, void *dst, int64_t count);

/* State import (checkpoint / hand-off from another implementation), in three steps:
 *   htm_import_begin(h, step_index)   the state being imported is "after step_index steps"
 *   htm_write(h, field, src, count)   the arrays of htm_read, same element types
 *   htm_import_commit(...)            the scalars that go with them; rebuilds derived state
 * SP permanences are imported with htm_sp_set_permanence.
 * htm_import_begin(h, HTM_IMPORT_PREV_STATE): TemporalMemory.process(..., prev_state=X) (networks.py:92-93) -- only the
 * fields of the previous step's State are written (cell words, winner cells, MATCH_* / SEG_POTENTIAL / CELL_MAX_JITTER);
 * the segment store, the step index and the sticky capacity flags stay what they are (`segments` of the commit is ignored). */
#define HTM_IMPORT_PREV_STATE (-1)
int htm_import_begin(htm_handle *h, int64_t step_index);
int htm_write(htm_handle *h, int32_t field, const void *src, int64_t count);
int htm_import_commit(htm_handle *h, int32_t segments, int32_t matching_segments, int32_t winner_cells,
                      int32_t has_distal_state, int32_t has_winner_cells);

/* Per-kernel device time of the most recent htm_run: names[i] / total_ms[i] / launches[i] for
 * up to max_kernels kernels (HIP events on the handle's stream).  Only collected when
 * htm_profile(h, 1) was called before the run (profiled runs are slower). */
int htm_profile(htm_handle *h, int32_t enable);
int htm_profile_read(htm_handle *h, int32_t max_kernels, const char **names, double *total_ms,
                     int64_t *launches);

/* Device-clock timeline of the pipelined launches (diagnostic; handle created with the environment
 * variable BITHTM_TRACE=1, otherwise HTM_ERR_STATE).  dst receives 8 x 4096 x 2 values: for slot =
 * launch (0..2; 0..3 in the four-launch schedule) + 4 * step parity and block b, the 100 MHz device wall clock when the block
 * started and when it ended, 0 where that block did not run.  Each launch overwrites its slot,
 * so after a run the buffer holds the last two steps -- of those with an index below BITHTM_TRACE_UNTIL,
 * if that is set (the last step of a run looks ahead less than the steady state).  Returns the
 * number of values. */
#define HTM_TRACE_VALUES (8 * 4096 * 2)
int64_t htm_trace_read(htm_handle *h, uint64_t *dst, int64_t count);

#ifdef __cplusplus
}
#endif
#endif /* BITHTM_HIP_H */
