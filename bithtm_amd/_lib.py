"""ctypes binding of libbithtm_hip.so (include/bithtm_hip.h).  No fallback: if the HIP
library is missing or cannot be loaded the import fails loudly."""

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libbithtm_hip.so")
ABI_VERSION = 4


class HtmConfig(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_uint32), ("device", C.c_int32),
        ("input_dim", C.c_int32), ("column_dim", C.c_int32), ("cell_dim", C.c_int32),
        ("active_columns", C.c_int32), ("enable_sp", C.c_int32), ("enable_tm", C.c_int32),
        ("sp_permanence_threshold", C.c_double), ("sp_delta_on", C.c_double), ("sp_delta_off", C.c_double),
        ("boost_coefficient", C.c_float), ("duty_momentum", C.c_float), ("duty_increment", C.c_float),
        ("tm_learn_active", C.c_double), ("tm_learn_inactive", C.c_double),
        ("tm_punish_active", C.c_double), ("tm_punish_inactive", C.c_double),
        ("tm_learn_prune", C.c_int32), ("tm_punish_prune", C.c_int32),
        ("tm_permanence_initial", C.c_float), ("tm_permanence_threshold", C.c_float),
        ("segment_activation_threshold", C.c_int32), ("segment_matching_threshold", C.c_int32),
        ("segment_sampling_synapses", C.c_int32),
        ("segment_capacity", C.c_int32), ("segment_capacity_local", C.c_int32), ("segment_slots", C.c_int32),
        ("seed", C.c_uint32), ("shard_rank", C.c_int32), ("shard_world", C.c_int32), ("use_caller_stream", C.c_int32),
        ("stream", C.c_void_p),
    ]


class HtmInfo(C.Structure):
    _fields_ = [
        ("step_index", C.c_int64), ("segments", C.c_int32), ("local_segments", C.c_int32), ("matching_segments", C.c_int32),
        ("winner_cells", C.c_int32), ("active_cells", C.c_int32), ("has_distal_state", C.c_int32),
        ("has_winner_cells", C.c_int32), ("capacity_error", C.c_int32), ("words_per_row", C.c_int32),
        ("new_segment_requests", C.c_int32), ("recycled_segments", C.c_int32), ("appended_segments", C.c_int32),
        ("work_items", C.c_int32), ("select_fallbacks", C.c_int32), ("candidate_exact_steps", C.c_int32), ("hot_select_steps", C.c_int32),
        ("select_zoom_steps", C.c_int32),
    ]


# htm_field
F_ACTIVE_COLUMN, F_OVERLAPS, F_BOOSTED, F_DUTY_CYCLE, F_CELL_ACTIVATION, F_CELL_PREDICTION = 1, 2, 3, 4, 5, 6
F_WINNER_WORDS, F_BURSTING, F_WINNER_CELL, F_SEG_CELL, F_SEG_NSYN, F_SEG_PRESYN, F_SEG_PERM = 7, 8, 9, 10, 11, 12, 13
F_SEGCOUNT, F_SEG_POTENTIAL, F_MATCH_SEGMENT, F_MATCH_INFO, F_MATCH_JITTER, F_CELL_MAX_JITTER = 14, 15, 16, 17, 18, 19
F_SEG_GID = 20
PLAN_GRAPH, PLAN_PIPELINED, PLAN_LEAN, PLAN_SCAN_LARGE = 1, 2, 4, 8
SP_OVERLAP, SP_BOOST, SP_SELECT, SP_ACTIVE, SP_LEARN, SP_DUTY, SP_COMMIT = 1, 2, 3, 4, 5, 6, 7

EXPORTS = {
    "htm_abi_version": (C.c_int, []),
    "htm_create": (C.c_int, [C.POINTER(HtmConfig), C.POINTER(C.c_void_p)]),
    "htm_destroy": (None, [C.c_void_p]),
    "htm_last_error": (C.c_char_p, [C.c_void_p]),
    "htm_sp_set_permanence": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "htm_set_epsilon": (C.c_int, [C.c_void_p, C.c_float]),
    "htm_sp_get_permanence": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "htm_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "htm_sp_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "htm_sp_phase": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]),
    "htm_tm_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "htm_tm_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "htm_tm_scan": (C.c_int, [C.c_void_p, C.c_void_p]),
    "htm_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "htm_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "htm_run_plan": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "htm_bank_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "htm_shard_record_bytes": (C.c_int64, [C.c_void_p]),
    "htm_shard_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "htm_shard_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "htm_shard_unique_id": (C.c_int, [C.c_void_p]),
    "htm_shard_comm_init": (C.c_int, [C.c_void_p, C.c_void_p]),
    "htm_shard_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]),
    "htm_shard_comm_size": (C.c_int, [C.c_void_p]),
    "htm_shard_graph_ok": (C.c_int, [C.c_void_p]),
    "htm_shard_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "htm_shard_group_run": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "htm_rccl_selftest": (C.c_int, [C.c_int32]),
    "htm_keyed_draws": (C.c_int, [C.c_uint32, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "htm_shard_group_step": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_int32]),
    "htm_populate": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_uint32]),
    "htm_sync": (C.c_int, [C.c_void_p]),
    "htm_get_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "htm_get_info": (C.c_int, [C.c_void_p, C.POINTER(HtmInfo)]),
    "htm_read": (C.c_int64, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]),
    "htm_read_rows": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_int64]),
    "htm_import_begin": (C.c_int, [C.c_void_p, C.c_int64]),
    "htm_write": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]),
    "htm_import_commit": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "htm_profile": (C.c_int, [C.c_void_p, C.c_int32]),
    "htm_profile_read": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                   C.POINTER(C.c_int64)]),
    "htm_trace_read": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64]),
}

_lib = None


def load():
    """Load the shared library once; raise ImportError (never fall back) if that fails."""
    global _lib
    if _lib is not None:
        return _lib
    # BITHTM_LIBRARY: another build of THIS library (same sources, same C ABI) -- the sanitizer build of its host code that
    # tests/test_host_sanitizers.py links against a HIP runtime made of host memory.  Not a fallback: unset, a missing
    # library is an error.
    path = os.environ.get("BITHTM_LIBRARY") or LIB_PATH
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP extension is not built. Run `python -m bithtm_amd.build` "
            "(or __graft_entry__.build()). There is no CPU fallback.")
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise ImportError(f"cannot load {path}: {e}. There is no CPU fallback.") from e
    for name, (restype, argtypes) in EXPORTS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.htm_abi_version() != ABI_VERSION:
        raise ImportError(f"{path}: ABI version {lib.htm_abi_version()} != {ABI_VERSION}; rebuild")
    _lib = lib
    return lib
