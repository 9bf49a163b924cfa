"""ctypes loader of libhispmv.so (the C ABI declared in include/hispmv.h).

There is no fallback: if the HIP library has not been built (``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C hispmv_amd/csrc``) importing this module
raises, and every compute entry point fails without a gfx950 device.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("HISPMV_LIB", _HERE / "lib" / "libhispmv.so"))

HISPMV_OK = 0
HISPMV_FULL = -1
HISPMV_EINVAL = -2
HISPMV_EDEVICE = -3
HISPMV_ESTATE = -4
HISPMV_ENOTDENSE = -5
HISPMV_EIO = -6
HISPMV_ENOMEM = -7


class MatrixInfo(C.Structure):
    _fields_ = [
        ("rows", C.c_int32), ("cols", C.c_int32), ("nnz", C.c_int64),
        ("is_dense", C.c_int32), ("loaded", C.c_int32),
        ("n_slices", C.c_int64), ("n_elems", C.c_int64), ("n_split_rows", C.c_int64),
        ("device_bytes", C.c_int64), ("prep_seconds", C.c_double),
        ("block_threads", C.c_int32), ("group_slices", C.c_int32), ("lds_bytes", C.c_int32), ("col_tiles", C.c_int32),
        ("carry_lookback", C.c_int32), ("col_tile_width", C.c_int32),
        ("col_tile_base", C.c_int32), ("compact_slices", C.c_int32),
        ("format", C.c_int32), ("tts_lines_per_gather", C.c_float), ("tile_kind", C.c_int32),
        ("batch_group_slices", C.c_int32),
    ]


_p = C.c_void_p
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)
_f32p = C.POINTER(C.c_float)

# name -> (restype, argtypes); every symbol include/hispmv.h declares.
SIGNATURES = {
    "hispmv_prep_build_tts": (C.c_int, [_p, C.c_int64, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "hispmv_prep_tts_array": (C.c_void_p, [_p, C.c_int]),
    "hispmv_prep_tts_pieces": (C.c_int, [_p, C.POINTER(C.c_int64)]),
    "hispmv_version": (C.c_char_p, []),
    "hispmv_host_threads": (C.c_int, []),
    "hispmv_free_failures": (C.c_int64, []),
    "hispmv_boundary_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "hispmv_boundary_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "hispmv_create": (C.c_int, [C.POINTER(_p), C.c_char_p] + [C.c_int] * 9),
    "hispmv_destroy": (None, [_p]),
    "hispmv_last_error": (C.c_char_p, [_p]),
    "hispmv_set_arena_bytes": (C.c_int, [_p, C.c_int64]),
    "hispmv_arena_bytes_used": (C.c_int64, [_p]),
    "hispmv_create_sparse_handle": (C.c_int, [_p, _p, _p, _p, C.c_int64, C.c_int32, C.c_int32]),
    "hispmv_create_sparse_handle_from_mtx": (C.c_int, [_p, C.c_char_p, C.c_int]),
    "hispmv_create_sparse_handle_from_csr": (C.c_int, [_p, _p, _p, _p, C.c_int32, C.c_int32]),
    "hispmv_create_dense_handle": (C.c_int, [_p, _p, C.c_int32, C.c_int32]),
    "hispmv_load_matrices": (C.c_int, [_p]),
    "hispmv_select_matrix": (C.c_int, [_p, C.c_uint32]),
    "hispmv_run_kernel": (C.c_int, [_p, _p, _p, _p, C.c_float, C.c_float]),
    "hispmv_linear": (C.c_int, [_p, C.c_int, _p, C.c_int64, _p, _p]),
    "hispmv_spmv_device": (C.c_int, [_p, C.c_int, _p, _p, _p, C.c_float, C.c_float, _p]),
    "hispmv_synchronize": (C.c_int, [_p]),
    "hispmv_last_kernel_ms": (C.c_float, [_p]),
    "hispmv_batch_graph_stats": (C.c_int, [_p, C.POINTER(C.c_int64)]),
    "hispmv_batch_call_info": (C.c_int, [_p, C.POINTER(C.c_int64)]),
    "hispmv_spmv_device_batch": (C.c_int, [_p, C.c_int32, _p, _p, _p, _p, C.c_float, C.c_float, _p]),
    "hispmv_time_device": (C.c_float, [_p, C.c_int, _p, _p, _p, C.c_float, C.c_float, C.c_int]),
    "hispmv_get_matrix_info": (C.c_int, [_p, C.c_int, C.POINTER(MatrixInfo)]),
    "hispmv_num_matrices": (C.c_int, [_p]),
    "hispmv_prep_from_coo": (C.c_int, [C.POINTER(_p), _p, _p, _p, C.c_int64, C.c_int32, C.c_int32]),
    "hispmv_prep_from_coo_device": (C.c_int, [C.POINTER(_p), C.c_int, _p, _p, _p, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "hispmv_prep_from_mtx": (C.c_int, [C.POINTER(_p), C.c_char_p, C.c_int]),
    "hispmv_prep_free": (None, [_p]),
    "hispmv_prep_last_error": (C.c_char_p, []),
    "hispmv_prep_dims": (C.c_int, [_p, _i64p]),
    "hispmv_prep_plan": (C.c_int, [_p, C.c_int, _i64p]),
    "hispmv_prep_choose_format": (C.c_int, [_p, C.c_int, _i64p]),
    "hispmv_prep_step_queue": (C.c_int, [_p, C.c_int32, _p, C.c_int32, C.c_int32, C.c_int32, _p, _p]),
    "hispmv_prep_window_membership": (C.c_int, [_p, C.c_int, C.c_void_p]),
    "hispmv_prep_apply_plan": (C.c_int, [_p, C.c_int, _i64p]),
    "hispmv_prep_groups": (_i32p, [_p]),
    "hispmv_prep_device_stream": (C.c_int, [_p, _i64p]),
    "hispmv_prep_device_array": (C.c_void_p, [_p, C.c_int]),
    "hispmv_prep_device_stream_on_device": (C.c_int, [_p, C.c_int, C.c_void_p, C.c_void_p]),
    "hispmv_prep_frags": (_i32p, [_p]),
    "hispmv_prep_csr_row_ptr": (_i64p, [_p]),
    "hispmv_prep_csr_col": (_i32p, [_p]),
    "hispmv_prep_csr_val": (_f32p, [_p]),
    "hispmv_prep_words": (_u64p, [_p]),
    "hispmv_prep_slice_hdr": (_i32p, [_p]),
    "hispmv_prep_fix": (_i32p, [_p]),
}


def _bind_single_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7, same as /opt/rocm's) and load it by file name, so a process that loads both
    copies ends up with two runtimes and the second one sees no GPU.  When torch is installed we
    therefore bind to ITS copy first (the loader then resolves our NEEDED libamdhip64.so.7 to it);
    otherwise the system ROCm runtime is used through our RUNPATH."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    cand = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if cand.exists():
        C.CDLL(str(cand), mode=C.RTLD_GLOBAL)


def _load() -> C.CDLL:
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: the HIP library is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C hispmv_amd/csrc`). "
            "There is no CPU fallback."
        )
    _bind_single_hip_runtime()
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()
