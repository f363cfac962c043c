"""ctypes mirror of include/pgmove.h. Fails loudly if libpgmove.so has not been built."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpgmove.so")

PG_OK = 0
PG_ERR_NO_DEVICE = -1
PG_ERR_INVALID_ARG = -2
PG_ERR_INPUT = -3
PG_ERR_RNA_FLAG = -4
PG_ERR_HIP = -5
PG_ERR_STATE = -6
PG_ERR_UNSUPPORTED = -7
PG_LOC_HOST = 0
PG_LOC_DEVICE = 1
PG_FLAG_LAZY_STATS = 1
PG_FLAG_PROFILE = 2
PG_FLAG_OVERLAP = 4
PG_FLAG_DEBUG_NARROW = 8
PG_FLAG_SHORT_READS_OK = 16
PG_FLAG_SKIP_OUT_OF_RANGE = 32
PG_FLAG_STOP_WHEN_FULL = 64
PG_FLAG_DEFER_STATS = 128
PG_FLAG_DEBUG_SPLIT_WALK = 256
PG_FLAG_OVERLAP_TAIL = 512
PG_FLAG_ONE_STREAM = 1024
PG_MODEL_KEEP_FIRST = 1
PG_MODEL_TEXT_MEDIAN, PG_MODEL_TEXT_SSTDEV, PG_MODEL_TEXT_DWELL = 0, 1, 2

# every symbol include/pgmove.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "pg_default_params", "pg_last_error", "pg_version", "pg_build_slot_tables", "pg_create", "pg_destroy",
    "pg_reset", "pg_submit", "pg_count", "pg_collect", "pg_stats", "pg_collect_gathered", "pg_job_totals_device", "pg_sync", "pg_finish", "pg_finish_deferred", "pg_fetch_samples", "pg_text", "pg_fetch_text", "pg_text_device", "pg_all_slots_full",
    "pg_last_batch_device", "pg_kernel_stats", "pg_kernel_stats_reset", "pg_set_stream", "pg_model", "pg_model_device", "pg_model_format",
    "pg_job_create", "pg_job_destroy", "pg_job_last_error", "pg_job_submit", "pg_job_submit_shards", "pg_job_reset", "pg_job_sync", "pg_job_all_slots_full", "pg_job_finish",
    "pg_job_finish_deferred", "pg_job_fetch_samples", "pg_job_text", "pg_job_fetch_text",
    "pg_job_uses_rccl", "pg_job_model", "pg_job_kernel_stats", "pg_runtime_init", "pg_all_slots_full_settled", "pg_job_all_slots_full_settled", "pg_poll", "pg_job_poll",
]
PG_JOB_EXCHANGE_AUTO, PG_JOB_EXCHANGE_HOST, PG_JOB_EXCHANGE_RCCL = 0, 1, 2


class PgParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("kmer_size", C.c_uint32), ("sig_move_offset", C.c_uint32),
        ("signal_print_margin", C.c_uint32), ("sample_limit", C.c_uint32), ("max_dur", C.c_uint32),
        ("min_dur", C.c_uint32), ("kmer_pick_margin", C.c_int32), ("scaling", C.c_int32),
        ("allow_rna", C.c_int32), ("pa_min", C.c_double), ("pa_max", C.c_double), ("n_slots", C.c_uint32),
        ("flags", C.c_uint32), ("device", C.c_int32), ("reserved", C.c_int32),
        ("table_t", C.c_void_p), ("table_u", C.c_void_p),
    ]


class PgBatch(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("location", C.c_int32), ("n_reads", C.c_uint32), ("n_ops", C.c_uint32),
        ("sig", C.c_void_p), ("sig_off", C.c_void_p), ("digitisation", C.c_void_p), ("offset", C.c_void_p),
        ("range", C.c_void_p), ("query_start", C.c_void_p), ("target_start", C.c_void_p), ("target_end", C.c_void_p),
        ("seq", C.c_void_p), ("seq_off", C.c_void_p), ("op_n", C.c_void_p), ("op_t", C.c_void_p), ("op_off", C.c_void_p),
        ("flags", C.c_uint32), ("reserved", C.c_uint32),
    ]


PG_BATCH_ALL_MATCHES = 1
PG_BATCH_RESIDENT = 2


class PgResult(C.Structure):
    _fields_ = [
        ("n_slots", C.c_uint32), ("reserved", C.c_uint32), ("n_events", C.c_uint64), ("n_samples", C.c_uint64),
        ("n_reads", C.c_uint64), ("counts", C.c_void_p), ("ev_off", C.c_void_p), ("ev_len", C.c_void_p),
        ("ev_read", C.c_void_p), ("samp_off", C.c_void_p), ("samples", C.c_void_p), ("read_skipped", C.c_void_p),
    ]


class PgTextResult(C.Structure):
    _fields_ = [("n_slots", C.c_uint32), ("reserved", C.c_uint32), ("n_bytes", C.c_uint64), ("slot_off", C.POINTER(C.c_uint64))]


class PgDeviceView(C.Structure):
    _fields_ = [
        ("n_events", C.c_uint64), ("n_samples", C.c_uint64), ("d_keep", C.c_void_p), ("d_ev_off", C.c_void_p),
        ("d_ev_len", C.c_void_p), ("d_ev_read", C.c_void_p), ("d_samp_off", C.c_void_p), ("d_samples", C.c_void_p),
        ("d_med", C.c_void_p), ("d_mad", C.c_void_p),
    ]


class PgModelResult(C.Structure):
    _fields_ = [
        ("n_slots", C.c_uint32), ("flags", C.c_uint32), ("n_values", C.c_void_p), ("median", C.c_void_p), ("sstdev", C.c_void_p),
        ("mid_lo", C.c_void_p), ("mid_hi", C.c_void_p), ("origin", C.c_void_p), ("sum1", C.c_void_p), ("sum2_lo", C.c_void_p),
        ("sum2_hi", C.c_void_p), ("dwell_n", C.c_void_p), ("dwell_median", C.c_void_p),
    ]


class PgKernelStat(C.Structure):
    _fields_ = [("name", C.c_char_p), ("launches", C.c_uint64), ("total_ms", C.c_double)]


_lib = None


def _preload_hip_runtime():
    """libpgmove.so has no DT_NEEDED on the HIP runtime (see Makefile): bring exactly one copy into the
    global symbol scope first. With PyTorch present that must be the libamdhip64 it bundles, otherwise the
    process would hold two HIP/HSA runtimes and the second one finds no GPU."""
    cands = []
    try:
        import torch  # noqa: F401  (loads its bundled ROCm libraries)
        cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:
        pass
    cands += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    last = None
    for path in cands:
        if os.path.isabs(path) and not os.path.exists(path):
            continue
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
            return path
        except OSError as e:  # try the next candidate
            last = e
    raise RuntimeError(f"no HIP runtime (libamdhip64) could be loaded: {last}")


def load():
    """Load libpgmove.so (once). Raises if the HIP library is missing: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make` (or __graft_entry__.build()). "
            "poregen_amd has no CPU fallback for the gmove path.")
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp, u32, i32, u64p = C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p
    lib.pg_default_params.argtypes = [C.POINTER(PgParams)]; lib.pg_default_params.restype = None
    lib.pg_last_error.argtypes = [vp]; lib.pg_last_error.restype = C.c_char_p
    lib.pg_version.argtypes = []; lib.pg_version.restype = C.c_char_p
    lib.pg_build_slot_tables.argtypes = [u32, C.POINTER(C.c_char_p), u32, vp, vp]; lib.pg_build_slot_tables.restype = i32
    lib.pg_create.argtypes = [C.POINTER(PgParams), C.POINTER(vp)]; lib.pg_create.restype = i32
    lib.pg_destroy.argtypes = [vp]; lib.pg_destroy.restype = None
    lib.pg_reset.argtypes = [vp]; lib.pg_reset.restype = i32
    lib.pg_submit.argtypes = [vp, C.POINTER(PgBatch)]; lib.pg_submit.restype = i32
    lib.pg_count.argtypes = [vp, C.POINTER(PgBatch), u64p, i32]; lib.pg_count.restype = i32
    lib.pg_collect.argtypes = [vp, u64p, i32]; lib.pg_collect.restype = i32
    lib.pg_stats.argtypes = [vp]; lib.pg_stats.restype = i32
    lib.pg_collect_gathered.argtypes = [vp, u64p, C.c_uint32, C.c_uint32]; lib.pg_collect_gathered.restype = i32
    lib.pg_finish_deferred.argtypes = [vp, C.POINTER(PgResult)]; lib.pg_finish_deferred.restype = i32
    lib.pg_fetch_samples.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_void_p]; lib.pg_fetch_samples.restype = i32
    lib.pg_text.argtypes = [vp, C.POINTER(PgTextResult)]; lib.pg_text.restype = i32
    lib.pg_fetch_text.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_void_p]; lib.pg_fetch_text.restype = i32
    lib.pg_job_totals_device.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]; lib.pg_job_totals_device.restype = i32
    lib.pg_sync.argtypes = [vp]; lib.pg_sync.restype = i32
    lib.pg_set_stream.argtypes = [vp, vp]; lib.pg_set_stream.restype = i32
    lib.pg_finish.argtypes = [vp, C.POINTER(PgResult)]; lib.pg_finish.restype = i32
    lib.pg_all_slots_full.argtypes = [vp]; lib.pg_all_slots_full.restype = i32
    lib.pg_last_batch_device.argtypes = [vp, C.POINTER(PgDeviceView)]; lib.pg_last_batch_device.restype = i32
    lib.pg_kernel_stats.argtypes = [vp, C.POINTER(PgKernelStat), u32, C.POINTER(u32)]; lib.pg_kernel_stats.restype = i32
    lib.pg_kernel_stats_reset.argtypes = [vp]; lib.pg_kernel_stats_reset.restype = i32
    lib.pg_model.argtypes = [vp, u32, C.POINTER(PgModelResult)]; lib.pg_model.restype = i32
    lib.pg_model_device.argtypes = [vp, u32, vp, vp, vp, vp, u32, C.POINTER(PgModelResult)]; lib.pg_model_device.restype = i32
    lib.pg_model_format.argtypes = [C.POINTER(PgModelResult), u32, i32, C.c_char_p, C.c_size_t]; lib.pg_model_format.restype = C.c_size_t
    lib.pg_runtime_init.argtypes = [i32]; lib.pg_runtime_init.restype = i32
    lib.pg_all_slots_full_settled.argtypes = [vp]; lib.pg_all_slots_full_settled.restype = i32
    lib.pg_job_all_slots_full_settled.argtypes = [vp]; lib.pg_job_all_slots_full_settled.restype = i32
    lib.pg_poll.argtypes = [vp]; lib.pg_poll.restype = i32
    lib.pg_job_poll.argtypes = [vp]; lib.pg_job_poll.restype = i32
    lib.pg_job_create.argtypes = [C.POINTER(PgParams), C.POINTER(C.c_int32), u32, u32, C.POINTER(vp)]; lib.pg_job_create.restype = i32
    lib.pg_job_destroy.argtypes = [vp]; lib.pg_job_destroy.restype = None
    lib.pg_job_last_error.argtypes = [vp]; lib.pg_job_last_error.restype = C.c_char_p
    lib.pg_job_submit.argtypes = [vp, C.POINTER(PgBatch)]; lib.pg_job_submit.restype = i32
    if hasattr(lib, "pg_job_submit_shards"):  # (a measurement build of an earlier round loaded through bench.py --lib lacks the newer entry points)
        lib.pg_job_submit_shards.argtypes = [vp, C.POINTER(PgBatch), C.c_uint32]; lib.pg_job_submit_shards.restype = i32
        lib.pg_job_reset.argtypes = [vp]; lib.pg_job_reset.restype = i32
    lib.pg_job_sync.argtypes = [vp]; lib.pg_job_sync.restype = i32
    lib.pg_job_all_slots_full.argtypes = [vp]; lib.pg_job_all_slots_full.restype = i32
    lib.pg_job_finish.argtypes = [vp, C.POINTER(PgResult)]; lib.pg_job_finish.restype = i32
    lib.pg_job_finish_deferred.argtypes = [vp, C.POINTER(PgResult)]; lib.pg_job_finish_deferred.restype = i32
    lib.pg_job_fetch_samples.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_void_p]; lib.pg_job_fetch_samples.restype = i32
    lib.pg_job_text.argtypes = [vp, C.POINTER(PgTextResult)]; lib.pg_job_text.restype = i32
    lib.pg_job_fetch_text.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_void_p]; lib.pg_job_fetch_text.restype = i32
    lib.pg_text_device.argtypes = [vp, u32, C.c_uint64, vp, vp, vp, C.POINTER(PgTextResult)]; lib.pg_text_device.restype = i32
    lib.pg_job_uses_rccl.argtypes = [vp]; lib.pg_job_uses_rccl.restype = i32
    lib.pg_job_model.argtypes = [vp, u32, C.POINTER(PgModelResult)]; lib.pg_job_model.restype = i32
    lib.pg_job_kernel_stats.argtypes = [vp, u32, C.POINTER(PgKernelStat), u32, C.POINTER(u32)]; lib.pg_job_kernel_stats.restype = i32
    _lib = lib
    return lib
