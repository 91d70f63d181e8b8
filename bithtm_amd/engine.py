"""One device engine = one handle of the C ABI (include/bithtm_hip.h): all device state of one
SpatialPooler and / or TemporalMemory.  Host logic only; every computation is a HIP kernel."""

import ctypes as C

import numpy as np

from . import _lib as L


class HtmError(RuntimeError):
    pass


class CapacityError(HtmError):
    """The fixed-capacity segment pool or a segment's synapse slots are exhausted."""


def pack_bits(bits, words):
    """bool[I] -> uint32[words], bit i of the input = bit (i & 31) of word (i >> 5).  An input that is packed already (a
    contiguous uint32 array of `words` words: a caller that keeps its inputs packed, or packs them once) goes through as it is."""
    if type(bits) is np.ndarray and bits.dtype == np.uint32 and bits.ndim == 1 and bits.size == words and bits.flags.c_contiguous:
        return bits
    bits = np.asarray(bits)
    if bits.dtype != np.bool_:
        bits = bits.astype(np.bool_)
    packed = np.packbits(bits, bitorder="little")
    if packed.size == words * 4:                    # (an input_dim that fills its words: no padding to add)
        return packed.view(np.uint32)
    out = np.zeros(words * 4, dtype=np.uint8)
    out[:len(packed)] = packed
    return out.view(np.uint32)


def words_per_column(cell_dim):
    """32-bit words of cells per column in the device's dense cell arrays: one up to 32 cells, two up to 64."""
    return max(1, -(-int(cell_dim) // 32))


def words_to_bool(words, cell_dim):
    """uint32[C * W] (W = words_per_column(cell_dim) words per column, side by side) -> bool[C, K]."""
    w = words_per_column(cell_dim)
    words = np.ascontiguousarray(words, dtype="<u4").reshape(-1, w)
    # (bit b of a little-endian word is bit b & 7 of its byte b >> 3: unpackbits in little bit order lays a column's cells out in
    # order -- a fifth of the time of shifting every word by 0..31)
    bits = np.unpackbits(words.view(np.uint8), axis=1, bitorder="little")
    return bits[:, :cell_dim].view(np.bool_) if cell_dim == 32 * w else np.ascontiguousarray(bits[:, :cell_dim]).view(np.bool_)


def bool_to_words(mat):
    """bool[C, K] -> uint32[C * W]."""
    mat = np.asarray(mat, dtype=np.bool_)
    C, K = mat.shape
    w = words_per_column(K)
    padded = np.zeros((C, 32 * w), dtype=np.uint32)
    padded[:, :K] = mat
    return (padded.reshape(C, w, 32) << np.arange(32, dtype=np.uint32)).sum(axis=2).astype(np.uint32).reshape(-1)


class Engine:
    def __init__(self, input_dim, column_dim, cell_dim, active_columns, proximal=None, boosting=None,
                 distal=None, seed=0, device=0, stream=None, shard_rank=0, shard_world=1):
        self.lib = L.load()
        self.input_dim = int(input_dim) if proximal is not None else 0
        self.column_dim = int(column_dim)
        self.cell_dim = int(cell_dim) if distal is not None else 0
        self.active_columns = int(active_columns)
        self.has_sp = proximal is not None
        self.has_tm = distal is not None
        self.cell_words = self.column_dim * words_per_column(self.cell_dim)     # length of the dense cell-word fields (F_CELL_*, F_WINNER_WORDS)
        cfg = L.HtmConfig()
        cfg.struct_bytes = C.sizeof(L.HtmConfig)
        cfg.device = device
        cfg.input_dim, cfg.column_dim, cfg.cell_dim = self.input_dim, self.column_dim, self.cell_dim
        cfg.active_columns = self.active_columns
        cfg.enable_sp, cfg.enable_tm = int(self.has_sp), int(self.has_tm)
        if self.has_sp:
            inc, dec = proximal.permanence_increment, proximal.permanence_decrement
            cfg.sp_permanence_threshold = proximal.permanence_threshold
            cfg.sp_delta_on = 1.0 * (inc + dec) - dec                 # projections.py:24
            cfg.sp_delta_off = 0.0 * (inc + dec) - dec
            density = boosting.active_outputs / boosting.output_dim     # regularizations.py:9
            cfg.boost_coefficient = np.float32(-(boosting.intensity / density))   # :16
            cfg.duty_momentum = np.float32(boosting.momentum)                     # :20
            cfg.duty_increment = np.float32(1.0 - boosting.momentum)              # :21
        if self.has_tm:
            a, b = distal.permanence_increment, -distal.permanence_decrement      # projections.py:287
            pa, pb = -distal.permanence_punishment, 0.0                           # :292
            cfg.tm_learn_active, cfg.tm_learn_inactive = 1.0 * (a - b) + b, 0.0 * (a - b) + b   # :102
            cfg.tm_punish_active, cfg.tm_punish_inactive = 1.0 * (pa - pb) + pb, 0.0 * (pa - pb) + pb
            cfg.tm_learn_prune, cfg.tm_punish_prune = int(min(a, b) < 0), int(min(pa, pb) < 0)   # :105
            cfg.tm_permanence_initial = np.float32(distal.permanence_initial)
            cfg.tm_permanence_threshold = np.float32(distal.permanence_threshold)
            cfg.segment_activation_threshold = distal.segment_activation_threshold
            cfg.segment_matching_threshold = distal.segment_matching_threshold
            cfg.segment_sampling_synapses = distal.segment_sampling_synapses
            cap = distal.segment_capacity
            cfg.segment_capacity = int(cap) if cap is not None else max(4096, 512 * self.active_columns)
            self._auto_grow = cap is None           # a default-sized pool grows like the reference's arrays (networks._grow_if_needed)
            cfg.segment_slots = int(distal.segment_slots)
            cfg.segment_capacity_local = int(getattr(distal, "segment_capacity_local", None) or 0)
        cfg.seed = int(seed) & 0xFFFFFFFF
        cfg.shard_rank, cfg.shard_world = int(shard_rank), int(shard_world)
        # stream: None = a private stream; "default" = the device's default stream; else a hipStream_t of the caller
        cfg.use_caller_stream = int(stream is not None)
        cfg.stream = stream if (stream and stream != "default") else None
        self._stream_owner = None                   # (an engine created on another engine's stream keeps that engine alive)
        self.shard_rank, self.shard_world = int(shard_rank), max(int(shard_world), 1)
        per = self.column_dim // self.shard_world
        self.column_range = (self.shard_rank * per, (self.shard_rank + 1) * per)
        self.segment_capacity, self.segment_slots = cfg.segment_capacity, cfg.segment_slots
        self.seed = cfg.seed
        handle = C.c_void_p()
        rc = self.lib.htm_create(C.byref(cfg), C.byref(handle))
        if rc != 0:
            raise HtmError(f"htm_create failed ({rc}): {self.lib.htm_last_error(None).decode()}")
        self.h = handle
        self.steps = 0
        self._banks = []
        if self.has_sp:
            c0, c1 = self.column_range          # a sharded handle only ever reads its own rows
            self.set_permanence(proximal.permanence[c0:c1] if hasattr(type(proximal), "permanence") else proximal._permanence[c0:c1],
                                row_begin=c0)
            self.words = self.info().words_per_row

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            self.lib.htm_destroy(h)

    # ---- plumbing
    def _check(self, rc, what):
        if rc < 0:
            raise HtmError(f"{what} failed ({rc}): {self.lib.htm_last_error(self.h).decode()}")
        return rc

    def info(self):
        """htm_info of the last completed step (does not raise on a capacity overflow: see
        check_capacity)."""
        out = L.HtmInfo()
        rc = self.lib.htm_get_info(self.h, C.byref(out))
        if rc != -3:                      # HTM_ERR_CAPACITY still fills `out`
            self._check(rc, "htm_get_info")
        return out

    def check_capacity(self):
        info = self.info()
        if info.capacity_error:
            raise CapacityError(self.lib.htm_last_error(self.h).decode())
        return info

    def sync(self):
        self._check(self.lib.htm_sync(self.h), "htm_sync")

    def stream_handle(self):
        """The hipStream_t this engine enqueues on (an integer; 0 = the default stream)."""
        out = C.c_void_p()
        self._check(self.lib.htm_get_stream(self.h, C.byref(out)), "htm_get_stream")
        return out.value or 0

    def shard_run(self, device_bank, n_inputs, n_steps, learning=True, use_graph=True, pipeline=True):
        """n_steps column-sharded timesteps, the exchange (RCCL) and the loop inside the library (htm_shard_run)."""
        flags = (1 if use_graph else 0) | (0 if pipeline else 2)
        self._check(self.lib.htm_shard_run(self.h, C.c_void_p(device_bank), int(n_inputs), int(n_steps), int(bool(learning)), flags), "htm_shard_run")
        self.steps += n_steps

    def import_prev_state(self, prediction, activation, winner_flat, distal):
        """TemporalMemory.process(prev_state=X) (networks.py:92-93): X's fields become the handle's "previous step" --
        cell predictions and activations (bool [C, K]), winner cells (flat ids, or None), the PredictiveProjection.State
        (or None) -- while the segment store and the step index stay what they are."""
        K = self.cell_dim
        S = self.info().segments
        self._check(self.lib.htm_import_begin(self.h, -1), "htm_import_begin")     # HTM_IMPORT_PREV_STATE: store, step index, flags stay
        self.write(L.F_CELL_PREDICTION, bool_to_words(np.asarray(prediction).reshape(self.column_dim, K)), np.uint32)
        self.write(L.F_CELL_ACTIVATION, bool_to_words(np.asarray(activation).reshape(self.column_dim, K)), np.uint32)
        winners = np.zeros(0, np.int32) if winner_flat is None else np.asarray(winner_flat, dtype=np.int32)
        self.write(L.F_WINNER_CELL, winners, np.int32)
        M = 0
        if distal is not None:
            seg = np.asarray(distal.matching_segment, dtype=np.int32)
            M = len(seg)
            pot = np.zeros(S, dtype=np.int64)              # (the store may have grown since X: potentials of newer segments are not used)
            old = np.asarray(distal.segment_potential, dtype=np.int64)[:S]
            pot[:len(old)] = old
            minfo = (pot[seg].astype(np.uint32) | (np.asarray(distal.matching_segment_activation).astype(np.uint32) << 12)
                     | (np.asarray(distal.matching_segment_active).astype(np.uint32) << 31))
            self.write(L.F_SEG_POTENTIAL, pot, np.int32)
            self.write(L.F_MATCH_SEGMENT, seg, np.int32)
            self.write(L.F_MATCH_INFO, minfo, np.uint32)
            self.write(L.F_MATCH_JITTER, np.asarray(distal.matching_segment_jittered_potential, dtype=np.float32), np.float32)
            self.write(L.F_CELL_MAX_JITTER, np.asarray(distal.max_jittered_potential, dtype=np.float32).view(np.uint32), np.uint32)
        self._check(self.lib.htm_import_commit(self.h, int(S), int(M), len(winners), int(distal is not None), int(winner_flat is not None)),
                    "htm_import_commit")

    def set_epsilon(self, epsilon):
        """TemporalMemory.process(epsilon=) (networks.py:91): 0 < epsilon <= 1, compared as float32; stays until set again."""
        self._check(self.lib.htm_set_epsilon(self.h, C.c_float(float(epsilon))), "htm_set_epsilon")

    def read(self, field, dtype, count):
        out = np.empty(int(count), dtype=dtype)
        n = self._check(self.lib.htm_read(self.h, field, out.ctypes.data_as(C.c_void_p), out.size), "htm_read")
        return out[:n]

    def read_rows(self, field, dtype, row_begin, row_count):
        """Rows [row_begin, row_begin + row_count) of a per-segment field (htm_read_rows)."""
        per = self.segment_slots if field in (L.F_SEG_PRESYN, L.F_SEG_PERM) else 1
        out = np.empty(int(row_count) * per, dtype=dtype)
        n = self._check(self.lib.htm_read_rows(self.h, field, int(row_begin), int(row_count), out.ctypes.data_as(C.c_void_p), out.size),
                        "htm_read_rows")
        return out[:n].reshape(int(row_count), per) if per > 1 else out[:n]

    def write(self, field, array, dtype):
        a = np.ascontiguousarray(array, dtype=dtype)
        self._check(self.lib.htm_write(self.h, field, a.ctypes.data_as(C.c_void_p), a.size), "htm_write")

    # ---- SP state
    def set_permanence(self, perm, row_begin=0):
        perm = np.ascontiguousarray(perm, dtype=np.float64)
        assert perm.ndim == 2 and perm.shape[1] == self.input_dim
        self._check(self.lib.htm_sp_set_permanence(self.h, perm.ctypes.data_as(C.c_void_p), row_begin, perm.shape[0]),
                    "htm_sp_set_permanence")

    def get_permanence(self, row_begin=0, row_count=None):
        row_count = self.column_dim - row_begin if row_count is None else row_count
        out = np.empty((row_count, self.input_dim), dtype=np.float64)
        self._check(self.lib.htm_sp_get_permanence(self.h, out.ctypes.data_as(C.c_void_p), row_begin, row_count),
                    "htm_sp_get_permanence")
        return out

    def read_duty_cycle(self):
        return self.read(L.F_DUTY_CYCLE, np.float32, self.column_dim)

    # ---- stepping
    def step(self, input_bits, learning=True):
        packed = pack_bits(input_bits, self.words)
        rc = self.lib.htm_step(self.h, packed.ctypes.data, 1 if learning else 0)
        if rc < 0:
            self._check(rc, "htm_step")
        self.steps += 1

    def sp_step(self, input_bits, learning=True):
        packed = pack_bits(input_bits, self.words)
        self._check(self.lib.htm_sp_step(self.h, packed.ctypes.data_as(C.c_void_p), int(bool(learning))), "htm_sp_step")
        self.steps += 1

    def sp_phase(self, phase, data=None, dtype=None):
        """One phase of SpatialPooler.process on the current timestep (htm_sp_phase, include/bithtm_hip.h)."""
        if data is None:
            rc = self.lib.htm_sp_phase(self.h, int(phase), None, 0)
        elif phase in (L.SP_OVERLAP, L.SP_LEARN):
            packed = pack_bits(data, self.words)
            rc = self.lib.htm_sp_phase(self.h, int(phase), packed.ctypes.data_as(C.c_void_p), packed.size)
        else:
            a = np.ascontiguousarray(data, dtype=dtype)
            rc = self.lib.htm_sp_phase(self.h, int(phase), a.ctypes.data_as(C.c_void_p), a.size)
        self._check(rc, "htm_sp_phase")
        if phase == L.SP_COMMIT:
            self.steps += 1

    def tm_step(self, active_column, learning=True, return_winner_cell=True):
        cols = np.ascontiguousarray(active_column, dtype=np.int32)
        self._check(self.lib.htm_tm_step(self.h, cols.ctypes.data_as(C.c_void_p), cols.size, int(bool(learning)),
                                         int(bool(return_winner_cell))), "htm_tm_step")
        self.steps += 1

    def tm_update(self, columns, winner_words, unaccounted_words, punish_words=None):
        """PredictiveProjection.update on its own (htm_tm_update, include/bithtm_hip.h)."""
        cols = np.ascontiguousarray(columns, dtype=np.int32)
        ww = np.ascontiguousarray(winner_words, dtype=np.uint32)
        uw = np.ascontiguousarray(unaccounted_words, dtype=np.uint32)
        pw = None if punish_words is None else np.ascontiguousarray(punish_words, dtype=np.uint32)
        self._check(self.lib.htm_tm_update(self.h, cols.ctypes.data_as(C.c_void_p), ww.ctypes.data_as(C.c_void_p), uw.ctypes.data_as(C.c_void_p),
                                           cols.size, None if pw is None else pw.ctypes.data_as(C.c_void_p)), "htm_tm_update")

    def tm_scan(self, active_words):
        """PredictiveProjection.process on its own (htm_tm_scan); closes the timestep."""
        aw = np.ascontiguousarray(active_words, dtype=np.uint32)
        assert aw.size == self.cell_words
        self._check(self.lib.htm_tm_scan(self.h, aw.ctypes.data_as(C.c_void_p)), "htm_tm_scan")
        self.steps += 1

    def upload_bank(self, inputs):
        """bool[n, I] -> device address of the packed bank htm_run reads."""
        inputs = np.asarray(inputs, dtype=np.bool_)
        n = inputs.shape[0]
        words = (self.input_dim + 31) // 32
        packed = np.zeros((n, words * 4), dtype=np.uint8)
        pb = np.packbits(inputs, axis=1, bitorder="little")
        packed[:, :pb.shape[1]] = pb
        ptr = C.c_void_p()
        self._check(self.lib.htm_bank_upload(self.h, packed.ctypes.data_as(C.c_void_p), n, C.byref(ptr)), "htm_bank_upload")
        return ptr.value

    def run(self, device_bank, n_inputs, n_steps, learning=True, use_graph=True, pipeline=True, continuing=False):
        """`continuing`: the next call is another run() on the same bank (HTM_RUN_CONTINUE, include/bithtm_hip.h)."""
        flags = (1 if use_graph else 0) | (0 if pipeline else 2) | (4 if continuing else 0)
        self._check(self.lib.htm_run(self.h, C.c_void_p(device_bank), int(n_inputs), int(n_steps), int(bool(learning)),
                                     flags), "htm_run")
        self.steps += n_steps

    def prepare(self, device_bank, n_inputs, n_steps, learning=True, use_graph=True, pipeline=True, continuing=False):
        """Build (capture + instantiate) the hipGraphs the run() call with these arguments will replay."""
        flags = (1 if use_graph else 0) | (0 if pipeline else 2) | (4 if continuing else 0)
        self._check(self.lib.htm_prepare(self.h, C.c_void_p(device_bank), int(n_inputs), int(n_steps), int(bool(learning)),
                                         flags), "htm_prepare")

    def run_plan(self, n_steps, use_graph=True, pipeline=True, continuing=False, **_):
        """What a run() with these arguments would do now (htm_run_plan): dict(hip_graph, pipelined, lean, scan_large)."""
        flags = (1 if use_graph else 0) | (0 if pipeline else 2) | (4 if continuing else 0)
        bits = self._check(self.lib.htm_run_plan(self.h, int(n_steps), flags), "htm_run_plan")
        return dict(hip_graph=bool(bits & L.PLAN_GRAPH), pipelined=bool(bits & L.PLAN_PIPELINED), lean=bool(bits & L.PLAN_LEAN),
                    scan_large=bool(bits & L.PLAN_SCAN_LARGE))

    # ---- column-sharded stepping (shard_world > 1): begin -> all-gather by the caller -> finish
    def shard_record_bytes(self):
        return int(self._check(self.lib.htm_shard_record_bytes(self.h), "htm_shard_record_bytes"))

    def shard_begin(self, send_ptr, input_bits=None, device_bank=None, n_inputs=1, learning=True):
        if input_bits is not None:
            packed = pack_bits(input_bits, self.words)
            rc = self.lib.htm_shard_begin(self.h, None, 1, packed.ctypes.data_as(C.c_void_p), int(bool(learning)), C.c_void_p(send_ptr))
        else:
            rc = self.lib.htm_shard_begin(self.h, C.c_void_p(device_bank), int(n_inputs), None, int(bool(learning)), C.c_void_p(send_ptr))
        self._check(rc, "htm_shard_begin")

    def shard_finish(self, recv_ptr, learning=True):
        self._check(self.lib.htm_shard_finish(self.h, C.c_void_p(recv_ptr), int(bool(learning))), "htm_shard_finish")
        self.steps += 1

    def shard_unique_id(self):
        """128-byte id for shard_comm_init (rank 0 creates it, the caller hands it to the other ranks)."""
        buf = C.create_string_buffer(128)
        rc = self.lib.htm_shard_unique_id(buf)
        if rc < 0:
            raise HtmError(f"htm_shard_unique_id failed ({rc}): {self.lib.htm_last_error(None).decode()}")
        return buf.raw

    def shard_comm_init(self, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self.lib.htm_shard_comm_init(self.h, buf), "htm_shard_comm_init")

    def shard_comm_size(self):
        return int(self._check(self.lib.htm_shard_comm_size(self.h), "htm_shard_comm_size"))

    def shard_graph_ok(self):
        """Whether shard_run replays whole timesteps, the all-gather included, as hipGraphs (htm_shard_comm_init's preflight)."""
        return bool(self._check(self.lib.htm_shard_graph_ok(self.h), "htm_shard_graph_ok"))

    def shard_step(self, input_bits=None, device_bank=None, n_inputs=1, learning=True):
        """One column-sharded timestep with the exchange (RCCL) inside the library."""
        if input_bits is not None:
            packed = pack_bits(input_bits, self.words)
            rc = self.lib.htm_shard_step(self.h, None, 1, packed.ctypes.data_as(C.c_void_p), int(bool(learning)))
        else:
            rc = self.lib.htm_shard_step(self.h, C.c_void_p(device_bank), int(n_inputs), None, int(bool(learning)))
        self._check(rc, "htm_shard_step")
        self.steps += 1

    def populate(self, segments_per_cell, synapses=32, perm_lo=0.3, perm_hi=0.7, seed=0, cell_begin=0, cell_end=None):
        """Pre-populated pool generated on the device (htm_populate; BASELINE.json configs[4])."""
        if cell_end is None:
            cell_end = self.column_dim * self.cell_dim
        self._check(self.lib.htm_populate(self.h, int(cell_begin), int(cell_end), int(segments_per_cell), int(synapses),
                                          float(perm_lo), float(perm_hi), int(seed) & 0xFFFFFFFF), "htm_populate")

    def profile(self, enable):
        self._check(self.lib.htm_profile(self.h, int(bool(enable))), "htm_profile")

    def profile_read(self, max_kernels=64):
        names = (C.c_char_p * max_kernels)()
        ms = (C.c_double * max_kernels)()
        cnt = (C.c_int64 * max_kernels)()
        n = self._check(self.lib.htm_profile_read(self.h, max_kernels, names, ms, cnt), "htm_profile_read")
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(n)}

    def trace_read(self):
        """Device-clock timeline of the pipelined launches (handle created under BITHTM_TRACE=1):
        int64 array [slot = launch + 4 * step parity][block][start, end], 100 MHz ticks, 0 = not run."""
        buf = np.zeros(8 * 4096 * 2, np.uint64)
        self._check(self.lib.htm_trace_read(self.h, buf.ctypes.data_as(C.c_void_p), buf.size), "htm_trace_read")
        return buf.astype(np.int64).reshape(8, 4096, 2)

    # ---- State fields (host views of the last completed step)
    def read_sp_fields(self):
        return dict(
            active_column=self.read(L.F_ACTIVE_COLUMN, np.int32, self.active_columns).astype(np.int64),
            overlaps=self.read(L.F_OVERLAPS, np.int32, self.column_dim).astype(np.int64),
            boosted_overlaps=self.read(L.F_BOOSTED, np.float64, self.column_dim))

    def read_store(self):
        """The segment store.  On a column-sharded handle the per-segment arrays have one row per LOCAL row (the
        segments of the rank's own cells); `seg_gid` gives each row's global id (-1 = free row)."""
        info = self.info()
        S, E = info.local_segments, self.segment_slots
        return dict(
            S=info.segments, rows=S, slots=E, seg_gid=self.read(L.F_SEG_GID, np.int32, S),
            seg_cell=self.read(L.F_SEG_CELL, np.int32, S), seg_nsyn=self.read(L.F_SEG_NSYN, np.int32, S),
            presyn=self.read(L.F_SEG_PRESYN, np.int32, S * E).reshape(S, E),
            perm=self.read(L.F_SEG_PERM, np.float32, S * E).reshape(S, E),
            segcount=self.read(L.F_SEGCOUNT, np.int32, self.column_dim * self.cell_dim))

    def read_distal(self):
        """PredictiveProjection.State (projections.py:195-203) of the last step, segment ids
        ascending as np.where (projections.py:247) yields them."""
        info = self.info()
        if not info.has_distal_state:
            return None
        S, M, N = info.local_segments, info.matching_segments, self.column_dim * self.cell_dim
        seg = self.read(L.F_MATCH_SEGMENT, np.int32, M).astype(np.int64)
        minfo = self.read(L.F_MATCH_INFO, np.uint32, M)
        jit = self.read(L.F_MATCH_JITTER, np.float32, M)
        order = np.argsort(seg, kind="stable")
        seg, minfo, jit = seg[order], minfo[order], jit[order]
        active = (minfo >> 31).astype(np.bool_)
        seg_cell = self.read(L.F_SEG_CELL, np.int32, S)
        prediction = np.bincount(seg_cell[seg], weights=active, minlength=N).astype(np.float64)
        return dict(
            prediction=prediction,
            segment_potential=self.read(L.F_SEG_POTENTIAL, np.int32, S).astype(np.int64),
            matching_segment=seg, matching_segment_activation=((minfo >> 12) & 0xFFF).astype(np.int64),
            matching_segment_active=active, max_jittered_potential=self.read(L.F_CELL_MAX_JITTER, np.float32, N),
            matching_segment_jittered_potential=jit)

    # ---- state hand-off in the oracle's dictionary layout (oracle/htm_oracle.py export_state)
    def export_tm_state(self):
        info = self.check_capacity()
        K = self.cell_dim
        st = self.read_store()
        out = dict(
            S=np.int64(st["S"]), slots=np.int64(st["slots"]), step_index=np.int64(info.step_index),
            seg_cell=st["seg_cell"], seg_nsyn=st["seg_nsyn"], presyn=st["presyn"], perm=st["perm"], segcount=st["segcount"],
            prev_prediction=words_to_bool(self.read(L.F_CELL_PREDICTION, np.uint32, self.cell_words), K),
            prev_activation=words_to_bool(self.read(L.F_CELL_ACTIVATION, np.uint32, self.cell_words), K),
            prev_winner=self.read(L.F_WINNER_CELL, np.int32, info.winner_cells).astype(np.int64),
            has_prev_winner=np.bool_(info.has_winner_cells), has_distal=np.bool_(info.has_distal_state))
        if self.shard_world > 1:                    # rows are local: distributed.merge_shard_states puts the ranks' parts together
            out.update(seg_gid=st["seg_gid"], column_range=np.asarray(self.column_range))
        d = self.read_distal()
        if d is not None:
            out.update(d)
        return out

    def import_tm_state(self, st):
        K, S, E = self.cell_dim, int(st["S"]), self.segment_slots
        if "seg_gid" in st:
            raise HtmError("import_tm_state: this is one rank's part of a sharded state; merge the parts first (distributed.merge_shard_states)")
        if S > self.segment_capacity:
            raise CapacityError(f"state has {S} segments, pool holds {self.segment_capacity}")
        self._check(self.lib.htm_import_begin(self.h, int(st["step_index"])), "htm_import_begin")
        presyn = np.asarray(st["presyn"], dtype=np.int32).reshape(S, -1) if S else np.zeros((0, E), np.int32)      # (an empty store: a checkpoint of step 0)
        perm = np.asarray(st["perm"], dtype=np.float32).reshape(S, -1) if S else np.zeros((0, E), np.float32)
        nsyn = (presyn >= 0).sum(axis=1).astype(np.int32)
        if nsyn.max(initial=0) > E:
            raise CapacityError(f"a segment has {nsyn.max()} synapses, segment_slots is {E}")
        # rows are packed on the device: valid synapses first
        order = np.argsort(presyn < 0, axis=1, kind="stable")[:, :E] if presyn.shape[1] else np.zeros((S, 0), np.int64)
        p_presyn = np.full((S, E), -1, dtype=np.int32)
        p_perm = np.full((S, E), -1.0, dtype=np.float32)
        w = min(E, presyn.shape[1])
        p_presyn[:, :w] = np.take_along_axis(presyn, order, axis=1)[:, :w]
        p_perm[:, :w] = np.take_along_axis(perm, order, axis=1)[:, :w]
        self.write(L.F_SEG_CELL, st["seg_cell"], np.int32)
        self.write(L.F_SEG_NSYN, nsyn, np.int32)
        self.write(L.F_SEG_PRESYN, p_presyn, np.int32)
        self.write(L.F_SEG_PERM, p_perm, np.float32)
        self.write(L.F_SEGCOUNT, st["segcount"], np.int32)
        self.write(L.F_CELL_PREDICTION, bool_to_words(np.asarray(st["prev_prediction"]).reshape(self.column_dim, K)), np.uint32)
        self.write(L.F_CELL_ACTIVATION, bool_to_words(np.asarray(st["prev_activation"]).reshape(self.column_dim, K)), np.uint32)
        winners = np.asarray(st["prev_winner"], dtype=np.int32)
        self.write(L.F_WINNER_CELL, winners, np.int32)
        M = 0
        has_distal = bool(st["has_distal"])
        if has_distal:
            seg = np.asarray(st["matching_segment"], dtype=np.int32)
            M = len(seg)
            pot = np.asarray(st["segment_potential"], dtype=np.int64)
            minfo = (pot[seg].astype(np.uint32) | (np.asarray(st["matching_segment_activation"]).astype(np.uint32) << 12)
                     | (np.asarray(st["matching_segment_active"]).astype(np.uint32) << 31))
            self.write(L.F_SEG_POTENTIAL, pot, np.int32)
            self.write(L.F_MATCH_SEGMENT, seg, np.int32)
            self.write(L.F_MATCH_INFO, minfo, np.uint32)
            self.write(L.F_MATCH_JITTER, st["matching_segment_jittered_potential"], np.float32)
            self.write(L.F_CELL_MAX_JITTER, np.asarray(st["max_jittered_potential"], dtype=np.float32).view(np.uint32), np.uint32)
        self._check(self.lib.htm_import_commit(self.h, S, M, len(winners), int(has_distal), int(bool(st["has_prev_winner"]))),
                    "htm_import_commit")
        self.steps = int(st["step_index"])
