"""Host-only access to the preprocessor (COO / MatrixMarket -> CSR -> slice stream) through the
C ABI's ``hispmv_prep_*`` entry points.  No device is needed; tests use it to compare the CSR
indices and the packed stream against the oracle on a CPU-only box.  Mirrors
``HiSpmvHandle::getPreparedMtx`` (common/src/spmv-helper.cpp:800-802)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from ._lib import lib, HISPMV_OK


@dataclass
class Prepared:
    rows: int
    cols: int
    nnz: int
    n_elems: int
    n_slices: int
    slice_elems: int
    stream_bytes: int
    row_ptr: np.ndarray    # int64 [rows+1]
    col_idx: np.ndarray    # int32 [nnz]
    values: np.ndarray     # float32 [nnz]
    words: np.ndarray      # uint64 [n_slices*slice_elems]: low 32 = fp32 bits, high 32 = rowEnd<<31 | col
    hdr: np.ndarray        # int32 [n_slices,4]: row_base, chain_len, x_base, x_span
    fix: np.ndarray        # int32 [n_split,4]: row, first_slice, len, 0
    plan: dict = None      # launch plan on a 256-CU device: threads, group_slices, lds_floats, ytile_floats, groups, lds_bytes
    staged_words: np.ndarray = None   # the stream as the device gets it: staged groups carry window indices, not columns
    groups: np.ndarray = None         # int32 [n_groups,4]: frag_begin, frag_count, lds_floats, 0
    frags: np.ndarray = None          # int32 [n_frags,4]: col_start, len, lds_off, 0
    tts: dict = None                  # the transposed tile stream (prep_from_coo(..., tts=True)): arrays + counts


def _collect_tts(p, target, geometry: int = 0):
    """geometry: 0 standard (8 K-row tiles), 1 small, "tall" = the list of the tall geometry's column parts (codes 2 + q)."""
    if isinstance(target, tuple):
        target, geometry = target
    if geometry in ("tall", "paired", "tallgap"):
        return [_collect_tts(p, target, {"tall": 2, "paired": 4, "tallgap": 8}[geometry] + q) for q in range(2)]
    geometry = 6 if geometry == "zerofill" else int(geometry)
    cnt = (C.c_int64 * 8)()
    lpg = C.c_double()
    if lib.hispmv_prep_build_tts(p, int(target), geometry, cnt, C.byref(lpg)) != HISPMV_OK:
        raise ValueError(lib.hispmv_prep_last_error().decode())
    tiles, blocks, slices, chunks, fillers, pads, max_rows, max_slots = (int(v) for v in cnt)

    def arr(which, n, dt, shape=None):
        ptr = lib.hispmv_prep_tts_array(p, which)
        if n == 0 or not ptr:
            a = np.zeros(0, dtype=dt)
        else:
            a = np.frombuffer((C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr), dtype=dt).copy()
        return a.reshape(shape) if shape else a
    pc = (C.c_int64 * 2)()
    lib.hispmv_prep_tts_pieces(p, pc)
    gap = geometry >= 8
    return dict(zero_fill=2 <= geometry < 8, flags_hi=arr(7, chunks * 64, np.uint16, (-1, 64)) if gap else None, n_tiles=tiles, n_blocks=blocks, n_slices=slices, n_chunks=chunks, fillers=fillers, pad_words=pads, max_rows=max_rows,
                max_slots=max_slots, lines_per_gather=float(lpg.value), n_carry=int(pc[1]), fix=arr(6, int(pc[0]) * 4, np.int32, (-1, 4)),
                words=arr(0, slices * 2048, np.uint32, (-1, 2, 1024)), col_base=arr(1, slices, np.int32), flags=arr(2, chunks * 64, np.uint16, (-1, 64)),
                chunk_info=arr(3, chunks * 2, np.int32, (-1, 2)), tiles=arr(4, tiles * 4, np.int32, (-1, 4)), blocks=arr(5, blocks * 8, np.int32, (-1, 8)))


def _collect(p, tts=None) -> Prepared:
    d = (C.c_int64 * 8)()
    lib.hispmv_prep_dims(p, d)
    rows, cols, nnz, n_elems, n_slices, se, n_fix, nbytes = (int(v) for v in d)

    def arr(ptr, n, shape=None):
        if n == 0:   # an empty std::vector hands out a NULL data pointer
            a = np.zeros(0, dtype=np.ctypeslib.as_ctypes_type(ptr._type_))
        else:
            a = np.ctypeslib.as_array(ptr, shape=(n,)).copy()
        return a.reshape(shape) if shape else a
    out = Prepared(rows, cols, nnz, n_elems, n_slices, se, nbytes,
                   arr(lib.hispmv_prep_csr_row_ptr(p), rows + 1),
                   arr(lib.hispmv_prep_csr_col(p), nnz),
                   arr(lib.hispmv_prep_csr_val(p), nnz),
                   arr(lib.hispmv_prep_words(p), n_slices * se),
                   arr(lib.hispmv_prep_slice_hdr(p), n_slices * 4, (-1, 4)),
                   arr(lib.hispmv_prep_fix(p), n_fix * 4, (-1, 4)))
    pl = (C.c_int64 * 6)()
    if lib.hispmv_prep_plan(p, 256, pl) == HISPMV_OK:
        out.plan = dict(zip(("threads", "group_slices", "lds_floats", "ytile_floats", "groups", "lds_bytes"), (int(v) for v in pl)))
    cnt = (C.c_int64 * 2)()
    if lib.hispmv_prep_apply_plan(p, 256, cnt) == HISPMV_OK:
        out.staged_words = arr(lib.hispmv_prep_words(p), n_slices * se)
        out.groups = arr(lib.hispmv_prep_groups(p), int(cnt[0]) * 4, (-1, 4))
        out.frags = arr(lib.hispmv_prep_frags(p), int(cnt[1]) * 4, (-1, 4))
    if tts is not None:
        out.tts = _collect_tts(p, tts)
    lib.hispmv_prep_free(p)
    return out


def prep_from_coo(coo_rows, coo_cols, coo_values, rows: int, cols: int, tts=None) -> Prepared:
    """tts: None, or the target elements per row tile of the transposed tile stream to pack as well (0 = loader's choice),
    or (target, geometry) with geometry 0 standard / 1 small / "tall" (Prepared.tts is then the list of the two column
    parts) -- see hispmv_prep_build_tts."""
    r = np.ascontiguousarray(coo_rows, dtype=np.int32)
    c = np.ascontiguousarray(coo_cols, dtype=np.int32)
    v = np.ascontiguousarray(coo_values, dtype=np.float32)
    p = C.c_void_p()
    rc = lib.hispmv_prep_from_coo(C.byref(p), C.c_void_p(r.ctypes.data), C.c_void_p(c.ctypes.data),
                                  C.c_void_p(v.ctypes.data), r.size, rows, cols)
    if rc != HISPMV_OK:
        raise ValueError(lib.hispmv_prep_last_error().decode())
    return _collect(p, tts)


def prep_from_coo_device(coo_rows, coo_cols, coo_values, rows: int, cols: int, device: int = 0):
    """The same object with both heavy stages computed on the MI355X (hispmv_prep_from_coo_device) -> (Prepared, seconds
    dict).  Needs a gfx950 device."""
    r = np.ascontiguousarray(coo_rows, dtype=np.int32)
    c = np.ascontiguousarray(coo_cols, dtype=np.int32)
    v = np.ascontiguousarray(coo_values, dtype=np.float32)
    p = C.c_void_p()
    secs = (C.c_double * 5)()
    rc = lib.hispmv_prep_from_coo_device(C.byref(p), int(device), C.c_void_p(r.ctypes.data), C.c_void_p(c.ctypes.data),
                                         C.c_void_p(v.ctypes.data), r.size, rows, cols, secs)
    if rc != HISPMV_OK:
        raise ValueError(lib.hispmv_prep_last_error().decode())
    return _collect(p), dict(zip(("upload", "csr_device", "offsets_host", "stream_device", "download"), (float(x) for x in secs)))


def prep_from_mtx(path, flavor: int = 0) -> Prepared:
    p = C.c_void_p()
    rc = lib.hispmv_prep_from_mtx(C.byref(p), str(path).encode(), int(flavor))
    if rc != HISPMV_OK:
        raise OSError(lib.hispmv_prep_last_error().decode())
    return _collect(p)


FORMAT_FIELDS = ("format", "tile_kind", "parts", "tile_width", "tile_base", "l2_tiles", "threads", "group", "lds_floats", "n_slices",
                 "n_elems", "n_split", "lines_per_gather_x1000", "l2_gather_elems")


def choose_format_from_csr(row_ptr, col_idx, values, rows: int, cols: int, n_cus: int = 256) -> dict:
    """The loader's format / tiling decision for a CSR matrix on an `n_cus`-CU device (hispmv_prep_choose_format): host-only,
    the same code path hispmv_create_sparse_handle_from_csr takes.  The CSR goes in as COO triplets in row-major order."""
    rp = np.asarray(row_ptr, dtype=np.int64)
    r = np.repeat(np.arange(rows, dtype=np.int32), np.diff(rp).astype(np.int64))
    c = np.ascontiguousarray(col_idx, dtype=np.int32)
    v = np.ascontiguousarray(values, dtype=np.float32)
    p = C.c_void_p()
    rc = lib.hispmv_prep_from_coo(C.byref(p), C.c_void_p(r.ctypes.data), C.c_void_p(c.ctypes.data), C.c_void_p(v.ctypes.data), r.size, rows, cols)
    if rc != HISPMV_OK:
        raise ValueError(lib.hispmv_prep_last_error().decode())
    try:
        out = (C.c_int64 * 16)()
        if lib.hispmv_prep_choose_format(p, int(n_cus), out) != HISPMV_OK:
            raise ValueError(lib.hispmv_prep_last_error().decode())
        return dict(zip(FORMAT_FIELDS, (int(x) for x in out)))
    finally:
        lib.hispmv_prep_free(p)


def window_membership(coo_rows, coo_cols, coo_values, rows: int, cols: int, n_cus: int = 256):
    """-> (inside, order): `inside[k]` for the matrix's CSR entries (1 = the entry's block of x lies in the LDS window of its
    workgroup, hispmv_prep_window_membership) and `order`, the permutation that brings the COO triplets into that CSR order."""
    r = np.ascontiguousarray(coo_rows, dtype=np.int32)
    c = np.ascontiguousarray(coo_cols, dtype=np.int32)
    v = np.ascontiguousarray(coo_values, dtype=np.float32)
    p = C.c_void_p()
    if lib.hispmv_prep_from_coo(C.byref(p), C.c_void_p(r.ctypes.data), C.c_void_p(c.ctypes.data), C.c_void_p(v.ctypes.data), r.size, rows, cols) != HISPMV_OK:
        raise ValueError(lib.hispmv_prep_last_error().decode())
    try:
        inside = np.zeros(r.size, dtype=np.uint8)
        if lib.hispmv_prep_window_membership(p, int(n_cus), C.c_void_p(inside.ctypes.data)) != HISPMV_OK:
            raise ValueError(lib.hispmv_prep_last_error().decode())
    finally:
        lib.hispmv_prep_free(p)
    return inside, np.lexsort((np.arange(r.size), c, r))     # stable: duplicates keep their input order, as in coo_to_csr


def device_layout_from_coo(coo_rows, coo_cols, coo_values, rows: int, cols: int, n_cus: int = 256, on_device: int = -1) -> dict:
    """The stream of a matrix as the device gets it (hispmv_prep_device_stream), host-only: -> dict with the host words before
    planning (`words`), the plan (`threads`, `group_slices`, `window_floats`, `stray_floats`, `groups`, `frags`) and the device
    arrays (`bytes` u8, `dgroups` [n,4], `stray_cols` [n_slices,64] or empty)."""
    r = np.ascontiguousarray(coo_rows, dtype=np.int32)
    c = np.ascontiguousarray(coo_cols, dtype=np.int32)
    v = np.ascontiguousarray(coo_values, dtype=np.float32)
    p = C.c_void_p()
    if lib.hispmv_prep_from_coo(C.byref(p), C.c_void_p(r.ctypes.data), C.c_void_p(c.ctypes.data), C.c_void_p(v.ctypes.data), r.size, rows, cols) != HISPMV_OK:
        raise ValueError(lib.hispmv_prep_last_error().decode())
    try:
        d = (C.c_int64 * 8)()
        lib.hispmv_prep_dims(p, d)
        n_slices, se = int(d[4]), int(d[5])
        words = np.ctypeslib.as_array(lib.hispmv_prep_words(p), shape=(n_slices * se,)).copy()
        pl = (C.c_int64 * 6)()
        lib.hispmv_prep_plan(p, int(n_cus), pl)
        cnt = (C.c_int64 * 2)()
        lib.hispmv_prep_apply_plan(p, int(n_cus), cnt)
        groups = np.ctypeslib.as_array(lib.hispmv_prep_groups(p), shape=(int(cnt[0]) * 4,)).copy().reshape(-1, 4) if cnt[0] else np.zeros((0, 4), np.int32)
        frags = np.ctypeslib.as_array(lib.hispmv_prep_frags(p), shape=(int(cnt[1]) * 4,)).copy().reshape(-1, 4) if cnt[1] else np.zeros((0, 4), np.int32)
        dc = (C.c_int64 * 6)()
        if lib.hispmv_prep_device_stream(p, dc) != HISPMV_OK:
            raise ValueError(lib.hispmv_prep_last_error().decode())

        def arr(which, n, dt):
            ptr = lib.hispmv_prep_device_array(p, which)
            return np.frombuffer((C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr), dtype=dt).copy() if (n and ptr) else np.zeros(0, dt)
        n_groups = int(dc[1])
        stray_floats = int(dc[4])
        extra = {}
        if on_device >= 0:          # the loader's layout kernel over the same planned words (needs a GPU): bytes_device / stray_cols_device
            bd = np.zeros(int(dc[0]), np.uint8)
            sd = np.zeros((n_slices, 64) if stray_floats else (0, 64), np.uint32)
            if lib.hispmv_prep_device_stream_on_device(p, int(on_device), C.c_void_p(bd.ctypes.data), C.c_void_p(sd.ctypes.data) if stray_floats else None) != HISPMV_OK:
                raise RuntimeError(lib.hispmv_prep_last_error().decode())
            extra = dict(bytes_device=bd, stray_cols_device=sd)
        return dict(**extra, words=words, n_slices=n_slices, threads=int(pl[0]), group_slices=int(pl[1]), window_floats=int(dc[5]), stray_floats=stray_floats,
                    compact_slices=int(dc[2]), stray_slices=int(dc[3]), groups=groups, frags=frags, bytes=arr(0, int(dc[0]), np.uint8),
                    dgroups=arr(1, n_groups * 4, np.int32).reshape(-1, 4),
                    stray_cols=arr(2, n_slices * 64 if stray_floats else 0, np.uint32).reshape(-1, 64))
    finally:
        lib.hispmv_prep_free(p)


def step_queue(slice_costs, tile_costs, n_wg: int = 256, mode: int = 0):
    """The order of the step kernel's queue (hispmv_prep_step_queue: the function the batch planner calls), host-only.
    -> (classes, indices): per queue position 0 = slice item / 1 = tile, and its index in that class's cost list."""
    a = np.ascontiguousarray(slice_costs, dtype=np.float64)
    b = np.ascontiguousarray(tile_costs, dtype=np.float64)
    n = a.size + b.size
    cls = np.zeros(n, np.int32)
    idx = np.zeros(n, np.int32)
    rc = lib.hispmv_prep_step_queue(C.c_void_p(a.ctypes.data) if a.size else None, a.size, C.c_void_p(b.ctypes.data) if b.size else None, b.size,
                                    int(n_wg), int(mode), C.c_void_p(cls.ctypes.data) if n else None, C.c_void_p(idx.ctypes.data) if n else None)
    if rc != HISPMV_OK:
        raise ValueError("hispmv_prep_step_queue: invalid argument")
    return cls, idx
