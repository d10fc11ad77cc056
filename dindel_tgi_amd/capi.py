"""ctypes binding of the C ABI in include/dindel_hmm.h (libdindel_hmm.so).

This is the test/bench harness' way into the product library — the same symbols a C++ host (the
reference's DetInDel::computeLikelihoods, DInDel.cpp:1707-1739) binds directly.  There is no Python
or CPU implementation behind it: if the shared library is missing, loading raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DD_LIB_PATH: A/B a diagnostic build of the SAME library (tools/); never a different implementation
LIB_PATH = os.environ.get("DD_LIB_PATH") or os.path.join(_HERE, "csrc", "libdindel_hmm.so")

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_u32p = C.POINTER(C.c_uint32)
c_u8p = C.POINTER(C.c_uint8)
c_i16p = C.POINTER(C.c_int16)
c_f64p = C.POINTER(C.c_double)

DD_TABLE_DOUBLES = 32 + 4 * 256 + 4 * 256 + 2 * 64 + 2 * 256 + 2 * 256 + 64
DD_READ_UNMAPPED, DD_READ_PAIRED, DD_READ_MATE_UNMAPPED, DD_READ_MATE_REVERSE, DD_READ_MATE_SAME_TID = 1, 2, 4, 8, 16

ABI_VERSION = 12           # DD_ABI_VERSION of include/dindel_hmm.h
DD_HPOS_INS, DD_HPOS_LO, DD_HPOS_RO, DD_HPOS_INS_KEY0 = -1, -3, -4, -16

DD_SUCCESS, DD_ERR_NO_DEVICE, DD_ERR_INVALID, DD_ERR_UNSUPPORTED, DD_ERR_HIP = 0, -1, -2, -3, -4
DD_PAIR_OK, DD_PAIR_HAPSIZE, DD_PAIR_NAN, DD_PAIR_LLPOS, DD_PAIR_UNSUPPORTED = 0, 1, 2, 3, 4
DD_MAX_HAP_LEN, DD_MAX_READ_LEN = 766, 1024


class dd_params(C.Structure):
    _fields_ = [("pError", C.c_double), ("pMut", C.c_double), ("pFirstgLO", C.c_double),
                ("mapQualThreshold", C.c_double), ("checkBaseQualThreshold", C.c_double),
                ("maxLengthDel", C.c_int32), ("padCover", C.c_int32), ("bMid", C.c_int32),
                ("forceReadOnHaplotype", C.c_int32), ("mapUnmappedReads", C.c_int32), ("maxMismatch", C.c_int32), ("capMapQualFast", C.c_double)]

    @classmethod
    def from_dict(cls, d):
        p = cls()
        for k, v in d.items():
            setattr(p, k, v)
        return p

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def params_cli_defaults():
    """DInDel.cpp:3937-3949 + 4122-4157 (the set production runs use)."""
    return dd_params(5e-4, 1e-5, 0.01, 100.0, 0.95, 5, 2, -1, 0, 0, 2, 45.0)


def params_struct_defaults():
    """ObservationModel.hpp:39-64."""
    return dd_params(1e-4, 1e-4, 0.01, 100.0, 0.95, 10, 5, -1, 0, 0, 1, 40.0)


class dd_batch(C.Structure):
    _fields_ = [("n_windows", C.c_int32),
                ("win_hap_off", c_i32p), ("win_read_off", c_i32p), ("win_hap_start", c_u32p),
                ("hap_seq_off", c_i32p), ("hap_seq", C.c_char_p), ("hap_var_off", c_i32p), ("hap_var", c_i32p),
                ("read_seq_off", c_i32p), ("read_seq", C.c_char_p), ("read_qidx", c_u8p), ("read_mqidx", c_u8p),
                ("read_start", c_u32p), ("read_flags", c_u8p),
                ("n_qual", C.c_int32), ("qual_table", c_f64p),
                ("n_mapq", C.c_int32), ("mapq_table", c_f64p), ("hap_var_flank", c_i32p),
                ("read_mate_pos", c_i32p), ("read_mate_len", c_i32p), ("read_lib", c_u8p),
                ("n_libs", C.c_int32), ("lib_off", c_i32p), ("lib_prob", c_f64p), ("lib_p95", c_f64p)]


RESULT_FIELDS = [("ll", c_f64p), ("llOn", c_f64p), ("llOff", c_f64p), ("mLogBQ", c_f64p),
                 ("offHap", c_u8p), ("offHapHMQ", c_u8p),
                 ("numIndels", c_i16p), ("numMismatch", c_i16p), ("nBQT", c_i16p), ("nmmBQT", c_i16p),
                 ("nMMLeft", c_i16p), ("nMMRight", c_i16p), ("firstBase", c_i16p), ("lastBase", c_i16p),
                 ("hpos", c_i16p), ("var_covered", c_u8p), ("status", c_i32p), ("onHap", c_u8p), ("var_fcov", c_u8p)]


class dd_result(C.Structure):
    _fields_ = RESULT_FIELDS


class dd_sizes(C.Structure):
    _fields_ = [("n_haps", C.c_int64), ("n_reads", C.c_int64), ("n_pairs", C.c_int64),
                ("hap_bases", C.c_int64), ("read_bases", C.c_int64), ("hpos_len", C.c_int64),
                ("var_cov_len", C.c_int64), ("cells", C.c_int64),
                ("max_hap_len", C.c_int32), ("max_read_len", C.c_int32)]


class dd_device_batch(C.Structure):
    _fields_ = [("n_windows", C.c_int32), ("n_haps", C.c_int32), ("n_reads", C.c_int32),
                ("max_hap_len", C.c_int32), ("max_read_len", C.c_int32), ("max_window_reads", C.c_int32),
                ("win_hap_off", C.c_void_p), ("win_read_off", C.c_void_p), ("win_hap_start", C.c_void_p),
                ("hap_seq_off", C.c_void_p), ("hap_seq", C.c_void_p), ("hap_var_off", C.c_void_p), ("hap_var", C.c_void_p),
                ("read_seq_off", C.c_void_p), ("read_seq", C.c_void_p), ("read_qidx", C.c_void_p), ("read_mqidx", C.c_void_p),
                ("read_start", C.c_void_p), ("read_flags", C.c_void_p),
                ("hap_window", C.c_void_p), ("win_pair_off", C.c_void_p), ("win_hpos_off", C.c_void_p),
                ("win_varcov_off", C.c_void_p), ("tables", C.c_void_p),
                ("n_qual", C.c_int32), ("n_mapq", C.c_int32), ("hap_var_flank", C.c_void_p), ("sym_lut", C.c_void_p),
                ("read_mate_pos", C.c_void_p), ("read_mate_len", C.c_void_p), ("read_lib", C.c_void_p),
                ("lib_off", C.c_void_p), ("lib_logprob", C.c_void_p), ("lib_log95", C.c_void_p),
                ("hap_class_list", C.c_void_p), ("classes", C.c_void_p), ("win_skip", C.c_void_p)]


# haplotype-length classes of a ragged batch, one per lane tiling (capi.cpp kHapClasses): longest haplotype, pairs per wavefront, positions per lane
HAP_CLASSES = [(30, 2, 1), (62, 1, 1), (94, 2, 3), (126, 1, 2), (158, 2, 5), (190, 1, 3), (222, 1, 4), (254, 1, 4), (318, 1, 5), (382, 1, 6),
               (446, 1, 7), (510, 1, 8), (574, 1, 9), (638, 1, 10), (702, 1, 11), (766, 1, 12)]
HAP_CLASS_BOUNDS = [c[0] for c in HAP_CLASSES]


N_HAP_CLASSES, N_READ_CLASSES = 16, 3


class dd_launch_class(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("list_off", "list_len", "hap_class", "max_hap_len", "min_read_len", "max_read_len",
                                         "max_window_reads", "avg_window_reads", "avg_read_len")]


class dd_length_classes(C.Structure):
    _fields_ = [("n_launches", C.c_int32), ("list_len", C.c_int32), ("launch", dd_launch_class * (N_HAP_CLASSES * N_READ_CLASSES))]


class dd_device_result(C.Structure):
    """dd_result with raw device addresses (same layout: every member is a pointer)."""
    _fields_ = [(n, C.c_void_p) for n, _ in RESULT_FIELDS]


EXPORTS = ["dd_params_struct_defaults", "dd_params_cli_defaults", "dd_batch_sizes", "dd_batch_offsets", "dd_screen_windows",
           "dd_compute_likelihoods", "dd_compute_likelihoods_faster", "dd_compute_likelihoods_multi", "dd_compute_likelihoods_faster_multi", "dd_partition_windows", "dd_launch_device_faster", "dd_release_cache", "dd_reserve_cache", "dd_host_alloc", "dd_host_free", "dd_build_tables", "dd_build_symbol_lut", "dd_build_library_tables", "dd_build_length_classes", "dd_plan_info", "dd_build_index", "dd_workspace_bytes",
           "dd_launch_device", "dd_kernel_name", "dd_last_launch", "dd_launch_log", "dd_last_direct_outputs", "dd_pair_sum_offsets", "dd_pair_sums_device",
           "dd_pair_sums", "dd_map_pairs_device", "dd_map_pairs", "dd_last_error", "dd_abi_version", "dd_device_count"]

_lib = None


def load():
    """Load libdindel_hmm.so; raises (loudly) when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the likelihood path)" % LIB_PATH)
    # One HIP runtime per process: torch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  Load it
    # FIRST so that our NEEDED libamdhip64.so.7 binds to the same copy; otherwise device pointers and
    # streams handed over from torch belong to a different runtime instance.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.dd_params_struct_defaults.argtypes = [C.POINTER(dd_params)]
    lib.dd_params_struct_defaults.restype = None
    lib.dd_params_cli_defaults.argtypes = [C.POINTER(dd_params)]
    lib.dd_params_cli_defaults.restype = None
    lib.dd_batch_sizes.argtypes = [C.POINTER(dd_batch), C.POINTER(dd_sizes)]
    lib.dd_batch_offsets.argtypes = [C.POINTER(dd_batch), c_i64p, c_i64p, c_i64p]
    lib.dd_compute_likelihoods.argtypes = [C.POINTER(dd_params), C.POINTER(dd_batch), C.POINTER(dd_result), C.c_int]
    lib.dd_compute_likelihoods_faster.argtypes = [C.POINTER(dd_params), C.POINTER(dd_batch), C.POINTER(dd_result), C.c_int]
    lib.dd_compute_likelihoods_multi.argtypes = [C.POINTER(dd_params), C.POINTER(dd_batch), C.POINTER(dd_result), c_i32p, C.c_int]
    lib.dd_compute_likelihoods_faster_multi.argtypes = lib.dd_compute_likelihoods_multi.argtypes
    lib.dd_partition_windows.argtypes = [C.POINTER(dd_batch), C.c_int, c_i32p]
    lib.dd_launch_device_faster.argtypes = [C.POINTER(dd_params), C.POINTER(dd_device_batch), C.POINTER(dd_device_result), C.c_void_p]
    lib.dd_release_cache.restype = None
    lib.dd_host_alloc.argtypes = [C.c_size_t]
    lib.dd_host_alloc.restype = C.c_void_p
    lib.dd_host_free.argtypes = [C.c_void_p]
    lib.dd_host_free.restype = None
    lib.dd_build_tables.argtypes = [C.POINTER(dd_params), c_f64p, C.c_int, c_f64p, C.c_int, c_f64p]
    lib.dd_build_symbol_lut.argtypes = [C.POINTER(dd_batch), C.POINTER(C.c_uint8)]
    lib.dd_build_library_tables.argtypes = [C.POINTER(dd_batch), c_f64p, c_f64p]
    lib.dd_build_length_classes.argtypes = [C.POINTER(dd_batch), c_u8p, C.POINTER(dd_params), c_i32p, C.POINTER(dd_length_classes)]
    lib.dd_screen_windows.argtypes = [C.POINTER(dd_batch), c_u8p, C.POINTER(C.c_int32 * 2)]
    lib.dd_plan_info.argtypes = [C.POINTER(dd_params), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32 * 10)]
    lib.dd_build_index.argtypes = [C.POINTER(dd_batch), c_i32p, c_i64p, c_i64p, c_i64p]
    lib.dd_workspace_bytes.argtypes = [C.POINTER(dd_params), C.POINTER(dd_device_batch)]
    lib.dd_workspace_bytes.restype = C.c_size_t
    lib.dd_launch_device.argtypes = [C.POINTER(dd_params), C.POINTER(dd_device_batch), C.POINTER(dd_device_result),
                                     C.c_void_p, C.c_size_t, C.c_void_p]
    lib.dd_pair_sum_offsets.argtypes = [C.POINTER(dd_batch), c_i64p]
    lib.dd_pair_sums_device.argtypes = [C.POINTER(dd_device_batch), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dd_pair_sums.argtypes = [C.POINTER(dd_batch), c_f64p, c_f64p, C.c_int]
    lib.dd_map_pairs_device.argtypes = [C.POINTER(dd_device_batch)] + [C.c_void_p] * 9
    lib.dd_map_pairs.argtypes = [C.POINTER(dd_batch), c_f64p, c_f64p, C.POINTER(C.c_uint8), c_i32p, c_f64p, c_f64p, c_i32p, c_f64p, C.c_int]
    lib.dd_last_launch.argtypes = [C.POINTER(C.c_int32 * 8)]
    lib.dd_last_launch.restype = None
    lib.dd_launch_log.argtypes = [c_i32p, C.c_int]
    lib.dd_last_direct_outputs.argtypes = []
    lib.dd_last_direct_outputs.restype = C.c_int
    lib.dd_kernel_name.restype = C.c_char_p
    lib.dd_last_error.restype = C.c_char_p
    lib.dd_abi_version.restype = C.c_int
    lib.dd_device_count.restype = C.c_int
    _lib = lib
    return lib


def last_launch():
    """Geometry of the last launch: dict(K, D, waves, lds_block, grid, split, lds_wave, lds_shared)."""
    out = (C.c_int32 * 8)()
    load().dd_last_launch(C.byref(out))
    return dict(zip(["K", "D", "waves", "lds_block", "grid", "split", "lds_wave", "lds_shared"], list(out)))


LAUNCH_LOG_FIELDS = ["K", "pairs_per_wave", "D", "gbt", "fold", "waves", "lds_block", "grid", "split", "n_haps", "max_hap", "min_read", "max_read",
                     "waves_per_cu", "occ", "us", "dynamic", "reads_per_wave"]


def launch_log():
    """Every main-model launch of the last dd_launch_device / dd_compute_likelihoods call of this thread (dd_launch_log)."""
    import numpy as np
    buf = np.zeros((64, len(LAUNCH_LOG_FIELDS)), np.int32)
    n = load().dd_launch_log(buf.ctypes.data_as(c_i32p), 64)
    return [dict(zip(LAUNCH_LOG_FIELDS, [int(v) for v in buf[i]])) for i in range(min(n, 64))]


def last_error():
    return load().dd_last_error().decode()


def hpos_reference_codes(hpos):
    """hpos as the library writes it (inserted bases carry their key: DD_HPOS_INS_KEY0 - pos) -> the reference's
    MLAlignment::hpos codes (every inserted base -1)."""
    import numpy as np
    h = np.asarray(hpos).copy()
    h[h < DD_HPOS_INS_KEY0] = DD_HPOS_INS
    return h


def kernel_source_id(kernel="dd_hmm_kernel"):
    """Identity of the sources a kernel is built from (12 hex digits of a SHA-1 over the files): profiles/*_pmc.json records it, and
    bench.py quotes a file's HBM-traffic counters only for the kernel they were measured on."""
    import hashlib
    import re
    files = ["hmm_kernel.h", "faster_kernel.hip" if "faster" in kernel else "hmm_kernel.hip"]
    h = hashlib.sha1()
    for f in files:                                # the code, not its comments or layout
        text = open(os.path.join(_HERE, "csrc", f), "r", errors="replace").read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        h.update(re.sub(r"\s+", " ", text).encode())
    return h.hexdigest()[:12]
