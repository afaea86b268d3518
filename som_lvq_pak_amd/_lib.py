"""ctypes loader for the C ABI in include/somhip.h (libsomhip.so, built in-tree by `make lib`).

There is no Python or CPU fallback: if the library is missing, or no MI355X is visible
when an engine is created, the call raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SOMHIP_LIB", os.path.join(HERE, "libsomhip.so"))   # override: A/B builds

c_float_p = C.POINTER(C.c_float)
c_i32_p = C.POINTER(C.c_int32)
c_i16_p = C.POINTER(C.c_int16)
c_u8_p = C.POINTER(C.c_uint8)
c_u64_p = C.POINTER(C.c_uint64)
c_i64_p = C.POINTER(C.c_int64)
c_double_p = C.POINTER(C.c_double)


class SomParams(C.Structure):
    _fields_ = [("length", C.c_int64), ("alpha", C.c_float), ("radius", C.c_float),
                ("alpha_type", C.c_int32), ("use_fixed", C.c_int32), ("use_weights", C.c_int32),
                ("batch", C.c_int64), ("start_iter", C.c_int64), ("count", C.c_int64),
                ("data_first", C.c_int64)]


class LvqParams(C.Structure):
    _fields_ = [("kind", C.c_int32), ("length", C.c_int64), ("alpha", C.c_float),
                ("alpha_type", C.c_int32), ("winlen", C.c_float), ("epsilon", C.c_float),
                ("start_iter", C.c_int64), ("count", C.c_int64), ("data_first", C.c_int64)]


# name -> (restype, argtypes); exactly the symbols include/somhip.h declares
SIGNATURES = {
    "somhip_last_error": (C.c_char_p, []),
    "somhip_version": (C.c_int, []),
    "somhip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "somhip_engine_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "somhip_engine_destroy": (None, [C.c_void_p]),
    "somhip_engine_stream": (C.c_void_p, [C.c_void_p]),
    "somhip_engine_sync": (C.c_int, [C.c_void_p]),
    "somhip_engine_set_scan_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "somhip_engine_set_update_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "somhip_scan_stats": (C.c_int, [C.c_void_p, c_u64_p]),
    "somhip_lvq_stats": (C.c_int, [C.c_void_p, c_u64_p]),
    "somhip_column_sums": (C.c_int, [C.c_void_p, c_float_p, C.POINTER(C.c_int64)]),
    "somhip_column_minmax": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.POINTER(C.c_int64)]),
    "somhip_centered_products": (C.c_int, [C.c_void_p, c_float_p, c_float_p]),
    "somhip_qerror2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int64, c_float_p, c_i32_p]),
    "somhip_debug_prefilter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, c_float_p, c_float_p, c_i64_p]),
    "somhip_codebook_create": (C.c_int, [C.c_void_p, c_float_p, c_i32_p, C.c_int64, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                         C.POINTER(C.c_void_p)]),
    "somhip_shard_units": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, c_i64_p, c_i64_p]),
    "somhip_codebook_create_interleaved": (C.c_int, [C.c_void_p, c_float_p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                                     C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "somhip_codebook_download": (C.c_int, [C.c_void_p, c_float_p]),
    "somhip_codebook_upload": (C.c_int, [C.c_void_p, c_float_p]),
    "somhip_codebook_destroy": (None, [C.c_void_p]),
    "somhip_dataset_create": (C.c_int, [C.c_void_p, c_float_p, C.c_int64, C.c_int, c_u8_p, c_i32_p,
                                        c_i16_p, c_i16_p, C.POINTER(C.c_void_p)]),
    "somhip_dataset_generate": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int64, C.c_int64, c_i32_p,
                                          C.POINTER(C.c_void_p)]),
    "somhip_dataset_wrap_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                             C.POINTER(C.c_void_p)]),
    "somhip_dataset_download_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, c_float_p]),
    "somhip_dataset_destroy": (None, [C.c_void_p]),
    "somhip_find_winners": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                      c_i32_p, c_float_p, c_i32_p]),
    "somhip_som_train": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(SomParams), c_i32_p, c_float_p]),
    "somhip_som_auto_batch": (C.c_int, [C.POINTER(SomParams), C.c_int64, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_int64),
                                        C.POINTER(C.c_int64)]),
    "somhip_lvq_train": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(LvqParams), c_float_p, c_i32_p,
                                   c_float_p]),
    "somhip_batch_winner_keys": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "somhip_shard_exchange_available": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "somhip_shard_winner_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "somhip_shard_winner_refine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "somhip_shard_winner_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "somhip_batch_topk_keys": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "somhip_merge_topk_keys": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p]),
    "somhip_lvq_rates_upload": (C.c_int, [C.c_void_p, c_float_p]),
    "somhip_lvq_rates_download": (C.c_int, [C.c_void_p, c_float_p]),
    "somhip_lvq_batch_candidates": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p]),
    "somhip_lvq_batch_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(LvqParams), C.c_int64, C.c_int64, C.c_int64,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_i64_p, c_i32_p, c_float_p]),
    "somhip_comm_unique_id": (C.c_int, [C.c_void_p]),
    "somhip_comm_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "somhip_comm_create_sockets": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "somhip_comm_allreduce_min_keys": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "somhip_comm_allreduce_sum_u32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "somhip_comm_allreduce_min_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "somhip_comm_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "somhip_comm_destroy": (None, [C.c_void_p]),
    "somhip_som_batch_update": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(SomParams), C.c_int64,
                                          C.c_int64, C.c_int64, C.c_void_p]),
    "somhip_device_alloc": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "somhip_device_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "somhip_copy_to_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "somhip_copy_to_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "somhip_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "somhip_timing_select": (C.c_int, [C.c_void_p, C.c_uint64]),
    "somhip_timing_reset": (C.c_int, [C.c_void_p]),
    "somhip_kernel_count": (C.c_int, []),
    "somhip_kernel_name": (C.c_char_p, [C.c_int]),
    "somhip_timing_get": (C.c_int, [C.c_void_p, C.c_int, c_i64_p, c_double_p]),
}

_lib = None


def load():
    """Load libsomhip.so (no GPU needed for loading; creating an engine needs one)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s not built -- run `make lib` (python -c 'import __graft_entry__ as g; "
                               "g.build()'); there is no CPU fallback" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class SomhipError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise SomhipError(load().somhip_last_error().decode())
