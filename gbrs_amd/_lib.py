"""ctypes binding of libgbrs_hip.so (include/gbrs_hip.h).

There is no CPU fallback: importing this module without the built library, or calling into
it without a visible gfx950 device, raises.  Build with ``python -c "import __graft_entry__ as g;
g.build()"``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GBRS_TUNING_LIB: a variant build of the same library (scripts/build_variant.sh), for A/B runs of kernel experiments
LIB_PATH = os.environ.get("GBRS_TUNING_LIB") or os.path.join(_HERE, "libgbrs_hip.so")

# names every build must export (checked by tests/test_abi.py against include/gbrs_hip.h)
EXPORTS = [
    "gbrs_last_error", "gbrs_abi_version", "gbrs_device_count", "gbrs_warm_up",
    "gbrs_em_create", "gbrs_em_create_device", "gbrs_em_create_masked", "gbrs_em_create_masked_device", "gbrs_em_set_initial_values", "gbrs_em_prepare", "gbrs_em_step", "gbrs_em_run",
    "gbrs_em_get", "gbrs_em_set_theta", "gbrs_em_group_sums", "gbrs_em_estep_partial",
    "gbrs_em_finish_step", "gbrs_em_prepare_partial", "gbrs_em_finish_prepare", "gbrs_em_stream",
    "gbrs_em_set_stream",
    "gbrs_em_sync", "gbrs_em_pair_begin", "gbrs_em_pair_check", "gbrs_em_pair_status", "gbrs_em_info", "gbrs_alignment_counts",
    "gbrs_counts_create", "gbrs_counts_get", "gbrs_counts_destroy", "gbrs_em_destroy",
    "gbrs_hmm_create", "gbrs_hmm_set_expression", "gbrs_hmm_set_eprob", "gbrs_hmm_run",
    "gbrs_hmm_get", "gbrs_hmm_info", "gbrs_hmm_destroy", "gbrs_interpolate", "gbrs_genoprob_dosage",
    "gbrs_compress_create", "gbrs_compress_get", "gbrs_compress_destroy",
    "gbrs_format_double", "gbrs_write_locus_table", "gbrs_parse_length_table", "gbrs_parse_genotype_table",
    "gbrs_decode_chunks", "gbrs_inflate_backend", "gbrs_zip_directory", "gbrs_npz_stack", "gbrs_zip_read_members", "gbrs_parse_number_table",
]

GBRS_OK = 0
GBRS_ERR_INVALID = -1
GBRS_ERR_HIP = -2
GBRS_ERR_NO_DEVICE = -3
GBRS_ERR_FLOAT = -4
GBRS_ERR_UNSUPPORTED = -5
GBRS_ERR_STATE = -6

GBRS_EM_DEFAULT = 0
GBRS_EM_MERGE_IDENTICAL_ROWS = 1
GBRS_EM_LAYOUT_CSC = 2
GBRS_EM_NO_INTERLEAVE = 4
GBRS_EM_FORCE_INTERLEAVE = 8
GBRS_EM_NO_STREAMS = 16
GBRS_EM_DETERMINISTIC = 32
GBRS_EM_KEEP_CSC = 64
GBRS_EM_SIDE_BY_SIDE = 128
GBRS_EM_ONE_SHOT = 256
GBRS_EM_NO_LOCUS_SETS = 512


class EmInfo(C.Structure):
    _fields_ = [
        ("num_rows", C.c_uint64), ("num_entries", C.c_uint64), ("num_device_rows", C.c_uint64),
        ("num_device_words", C.c_uint64), ("bytes_per_iter", C.c_uint64),
        ("algorithmic_bytes", C.c_uint64), ("last_estep_ms", C.c_double),
        ("last_step_ms", C.c_double), ("num_loci", C.c_uint32), ("num_haps", C.c_uint32),
        ("layout", C.c_uint32), ("reserved", C.c_uint32),
        ("num_tiles", C.c_uint64), ("num_slots", C.c_uint64), ("num_long_rows", C.c_uint64),
        ("num_heavy_loci", C.c_uint64), ("num_light_loci", C.c_uint64), ("estep_bytes", C.c_uint64),
        ("retained_build_bytes", C.c_uint64),
        ("num_locus_sets", C.c_uint64),
    ]


class HmmInfo(C.Structure):
    _fields_ = [
        ("total_genes", C.c_uint64), ("algorithmic_bytes", C.c_uint64),
        ("last_emission_ms", C.c_double), ("last_forward_ms", C.c_double),
        ("last_backward_ms", C.c_double), ("last_backtrace_ms", C.c_double),
        ("num_states", C.c_int32), ("n_samples", C.c_int32),
        ("last_run_ms", C.c_double),
        ("last_delta_blocks", C.c_int32), ("last_delta_longest_fixup", C.c_int32),
        ("last_delta_fallbacks", C.c_int32), ("reserved0", C.c_int32),
    ]


class GbrsHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


_lib = None


def load():
    """Load libgbrs_hip.so (once).  Raises ImportError with build instructions if missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP library has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "gbrs_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, u32, u64, i64, dbl = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_int64, C.c_double
    pp = C.POINTER(C.c_void_p)
    lib.gbrs_last_error.restype = C.c_char_p
    lib.gbrs_last_error.argtypes = []
    lib.gbrs_abi_version.restype = i32
    lib.gbrs_device_count.restype = i32
    lib.gbrs_warm_up.restype = i32
    lib.gbrs_warm_up.argtypes = [i32]
    sigs = {
        "gbrs_em_create": [u64, u32, u32, pp, pp, vp, vp, i32, u32, pp],
        "gbrs_em_create_device": [u64, u32, u32, pp, pp, vp, vp, i32, u32, pp],
        "gbrs_em_create_masked": [u64, u32, u32, pp, pp, vp, vp, vp, i32, u32, pp],
        "gbrs_em_create_masked_device": [u64, u32, u32, pp, pp, vp, vp, vp, i32, u32, pp],
        "gbrs_em_set_initial_values": [vp, pp],
        "gbrs_em_prepare": [vp, dbl],
        "gbrs_em_step": [vp, i32, C.POINTER(dbl)],
        "gbrs_em_run": [vp, i32, dbl, i32, C.POINTER(i32), vp, i32, vp],
        "gbrs_em_get": [vp, vp, vp],
        "gbrs_em_set_theta": [vp, vp],
        "gbrs_em_group_sums": [vp, i64, vp, vp, i32, vp],
        "gbrs_em_estep_partial": [vp, pp, C.POINTER(u64)],
        "gbrs_em_finish_step": [vp, C.POINTER(dbl)],
        "gbrs_em_prepare_partial": [vp, pp, C.POINTER(u64)],
        "gbrs_em_finish_prepare": [vp, dbl],
        "gbrs_em_sync": [vp],
        "gbrs_em_pair_begin": [vp, vp, i32],
        "gbrs_em_pair_check": [vp, vp, dbl],
        "gbrs_em_pair_status": [vp, vp, C.POINTER(i32), C.POINTER(i32), vp, i32],
        "gbrs_em_set_stream": [vp, vp],
        "gbrs_em_info": [vp, C.POINTER(EmInfo)],
        "gbrs_alignment_counts": [u64, u32, u32, pp, pp, vp, vp, u32, i32, vp, vp, vp],
        "gbrs_counts_create": [u64, u32, u32, pp, pp, vp, i32, pp],
        "gbrs_counts_get": [vp, vp, u32, vp, vp, vp],
        "gbrs_counts_destroy": [vp],
        "gbrs_em_destroy": [vp],
        "gbrs_hmm_create": [i32, i32, vp, vp, pp, i32, pp],
        "gbrs_hmm_set_expression": [vp, i32, pp, pp, pp, dbl, dbl],
        "gbrs_hmm_set_eprob": [vp, i32, pp],
        "gbrs_hmm_run": [vp],
        "gbrs_hmm_get": [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp],
        "gbrs_hmm_info": [vp, C.POINTER(HmmInfo)],
        "gbrs_hmm_destroy": [vp],
        "gbrs_interpolate": [i32, i32, vp, vp, i32, vp, vp, i32],
        "gbrs_genoprob_dosage": [i32, i64, vp, vp, i32],
        "gbrs_compress_create": [u64, u32, u32, pp, pp, vp, i32, pp, C.POINTER(u64), vp],
        "gbrs_compress_get": [vp, pp, pp, vp],
        "gbrs_compress_destroy": [vp],
    }
    sigs.update(_host_signatures())
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.restype = i32
        fn.argtypes = args
    lib.gbrs_em_stream.restype = vp
    lib.gbrs_em_stream.argtypes = [vp]
    if lib.gbrs_abi_version() != 5:
        raise ImportError("libgbrs_hip.so ABI version mismatch")
    _lib = lib
    return lib


def _host_signatures():
    """The host-only entry points of the library (gbrs_amd/csrc/hostio.hip)."""
    vp, i32, u32, u64, i64, dbl = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_int64, C.c_double
    return {
        "gbrs_format_double": [dbl, C.c_char_p],
        "gbrs_decode_chunks": [C.c_char_p, i64, vp, vp, vp, vp, u64, u32, u64, i32, i32, vp, i32],
        "gbrs_inflate_backend": [],
        "gbrs_zip_directory": [vp, u64, u64, vp, vp, vp, vp, vp, vp, u64, C.POINTER(u64), C.POINTER(u64)],
        "gbrs_npz_stack": [vp, u64, i64, vp, vp, vp, vp, vp, vp, u64, u64, vp, vp, i32],
        "gbrs_zip_read_members": [vp, u64, i64, vp, vp, vp, vp, vp, vp, i32],
        "gbrs_parse_number_table": [C.c_char_p, i64, i64, i32, vp],
        "gbrs_parse_length_table": [C.c_char_p, i64, C.c_char_p, vp, i64, C.c_char_p, vp, i32, dbl, vp],
        "gbrs_parse_genotype_table": [C.c_char_p, i64, C.c_char_p, vp, i64, C.c_char_p, vp, i32, vp, vp, i32, vp,
                                      C.POINTER(i64)],
        "gbrs_write_locus_table": [C.c_char_p, C.c_char_p, vp, i64, i32, i64, i64, vp, C.c_char_p, vp, C.c_char_p, vp, vp],
    }


def check(status):
    """Map a gbrs_status to the exception the reference would have raised."""
    if status == GBRS_OK:
        return
    msg = load().gbrs_last_error().decode(errors="replace")
    if status == GBRS_ERR_FLOAT:
        raise FloatingPointError(msg)
    raise GbrsHipError(status, msg)


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def ptr_table(arrays):
    """host array of H pointers to the given numpy arrays (kept alive by the caller)."""
    tab = (C.c_void_p * len(arrays))()
    for i, a in enumerate(arrays):
        tab[i] = None if a is None else a.ctypes.data
    return tab


def raw_table(addresses):
    tab = (C.c_void_p * len(addresses))()
    for i, a in enumerate(addresses):
        tab[i] = a
    return tab


def warm_up_device_async(device=0):
    """Start the HIP runtime (library load, hipInit, device context: ~0.15-0.3 s in a fresh process) on a
    background thread so that it overlaps with reading the input files; ctypes releases the GIL during
    the call.  Returns the thread (join() is optional: the first real call blocks on the runtime's own
    initialisation lock anyway)."""
    import threading

    def _go():
        try:
            load().gbrs_warm_up(int(device))
        except Exception:      # noqa: BLE001 - the foreground call will report the problem
            pass
    t = threading.Thread(target=_go, name='gbrs-hip-init', daemon=True)
    t.start()
    return t
