"""oracle -- TEST INFRASTRUCTURE ONLY.

ctypes access to ``liboracle.so`` (CPU restatement of the reference's SpMV path, see
hispmv_oracle.cpp) and, when it has been built in this container, ``_ref/libref_cpu.so``
(the reference's own cpu/ MatrixMarket loader compiled from /root/reference).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package -- as the checker, never as the thing measured or shipped.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "liboracle.so"
REF_PATH = _HERE / "_ref" / "libref_cpu.so"

_p = C.c_void_p
_ip = C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)


def build(force: bool = False) -> None:
    """Compile liboracle.so (and _ref/ when /root/reference is present)."""
    if force or not LIB_PATH.exists():
        subprocess.run(["make", "-C", str(_HERE), "-B" if force else "-s", "all"], check=True,
                       stdout=subprocess.DEVNULL)


def _load():
    if not LIB_PATH.exists():
        build()
    lib = C.CDLL(str(LIB_PATH))
    lib.orc_free.argtypes = [_p]
    lib.orc_read_mtx_cpu.argtypes = [C.c_char_p, _ip, _ip, _ip, C.POINTER(_ip), C.POINTER(_ip), C.POINTER(_fp)]
    lib.orc_load_mtx_common.argtypes = lib.orc_read_mtx_cpu.argtypes
    lib.orc_cpu_spmv.argtypes = [C.c_int, _p, _p, _p, _p, _p, C.c_float, C.c_float, C.c_int]
    lib.orc_cpu_sequential.argtypes = [C.c_int64, _p, _p, _p, C.c_int, _p, _p, C.c_float, C.c_float, _p]
    lib.orc_naive_gemv.argtypes = [C.c_int, C.c_int, _p, _p, _p, C.c_float, C.c_float, C.c_int]
    lib.orc_precision_loss.argtypes = [C.c_int64, _p, _p, C.POINTER(C.c_double), _ip]
    lib.orc_precision_loss.restype = C.c_double
    lib.orc_spmv_f64.argtypes = [C.c_int, _p, _p, _p, _p, _p, C.c_float, C.c_float, _p, _p]
    lib.orc_omp_spmv_timed.argtypes = [C.c_int, _p, _p, _p, _p, _p, _p, C.c_float, C.c_float, C.c_int, _ip]
    lib.orc_omp_spmv_timed.restype = C.c_double
    lib.orc_mkl_available.restype = C.c_int
    lib.orc_mkl_spmv.argtypes = [C.c_int, C.c_int, _p, _p, _p, _p, _p, C.c_float, C.c_float, C.c_int, C.c_int, _ip]
    lib.orc_mkl_spmv.restype = C.c_double
    lib.orc_cpu_bench_spmv.argtypes = [C.c_int, C.c_int, _p, _p, _p, C.c_int, C.c_int, C.c_double, C.c_int, _ip]
    lib.orc_cpu_bench_spmv.restype = C.c_double
    lib.orc_cpu_bench_gemv.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, _ip]
    lib.orc_cpu_bench_gemv.restype = C.c_double
    lib.orc_print_error_stats.argtypes = [C.c_int64, _p, C.c_int64, _p, C.c_char_p, C.c_int]
    lib.orc_print_error_stats.restype = C.c_int
    lib.refpack_create.argtypes = [C.c_int] * 8
    lib.refpack_create.restype = _p
    lib.refpack_free.argtypes = [_p]
    lib.refpack_prepare_sparse.argtypes = [_p, C.c_int, C.c_int, C.c_int64, _p, _p, _p]
    lib.refpack_prepare_dense.argtypes = [_p, C.c_int, C.c_int, _p]
    lib.refpack_info.argtypes = [_p, _p]
    lib.refpack_channel.argtypes = [_p, C.c_int]
    lib.refpack_channel.restype = C.POINTER(C.c_uint64)
    lib.refpack_hash.argtypes = [_p]
    lib.refpack_hash.restype = C.c_uint64
    lib.refpack_count_bit.argtypes = [_p, C.c_int]
    lib.refpack_count_bit.restype = C.c_int64
    lib.refpack_tile_size.argtypes = [_p, C.c_int, C.c_int]
    lib.refpack_shared_rows.argtypes = [_p, C.c_int, C.c_int, _p, C.c_int]
    lib.refpack_emulate.argtypes = [_p, _p, _p, C.c_float, C.c_float, _p]
    lib.emu_spmv.argtypes = [_p, _p, _p, C.c_int64, C.c_int64, _p, _p, C.c_float, C.c_float, _p, C.c_int32, C.c_int]
    lib.emu_gemv.argtypes = [_p, C.c_int32, C.c_int32, _p, _p, C.c_float, C.c_float, _p]
    lib.emu_tts.argtypes = [_p, _p, _p, _p, _p, _p, C.c_int64, _p, C.c_int64, C.c_int64, _p, _p, C.c_float, C.c_float, _p, C.c_int, _p]
    return lib


lib = _load()


def _ptr(a: np.ndarray):
    return C.c_void_p(a.ctypes.data)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _take(ptr, n, dt):
    out = np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].astype(dt, copy=True)
    lib.orc_free(C.cast(ptr, _p))
    return out


def _mtx_call(fn, path):
    rows, cols, nnz = C.c_int(), C.c_int(), C.c_int()
    a, b, v = _ip(), _ip(), _fp()
    rc = fn(str(path).encode(), C.byref(rows), C.byref(cols), C.byref(nnz), C.byref(a), C.byref(b), C.byref(v))
    if rc != 0:
        raise OSError(f"oracle loader failed with {rc} on {path}")
    return rows.value, cols.value, nnz.value, a, b, v


def read_mtx_cpu(path):
    """cpu/ loader restatement -> (rows, cols, row_ptr, col_idx, vals) (CSR)."""
    rows, cols, nnz, a, b, v = _mtx_call(lib.orc_read_mtx_cpu, path)
    return rows, cols, _take(a, rows + 1, np.int32), _take(b, nnz, np.int32), _take(v, nnz, np.float32)


def load_mtx_common(path):
    """common/ loadMtx restatement -> (rows, cols, coo_r, coo_c, coo_v) in file order."""
    rows, cols, nnz, a, b, v = _mtx_call(lib.orc_load_mtx_common, path)
    return rows, cols, _take(a, nnz, np.int32), _take(b, nnz, np.int32), _take(v, nnz, np.float32)


def ref_available() -> bool:
    return REF_PATH.exists()


def ref_read_mtx_csr(path):
    """The REFERENCE's readMatrixCSC + convertCSCtoCSR (oracle/_ref, built in this container)."""
    ref = C.CDLL(str(REF_PATH))
    ref.ref_read_mtx_csr.argtypes = lib.orc_read_mtx_cpu.argtypes
    ref.ref_free.argtypes = [_p]
    rows, cols, nnz = C.c_int(), C.c_int(), C.c_int()
    a, b, v = _ip(), _ip(), _fp()
    ref.ref_read_mtx_csr(str(path).encode(), C.byref(rows), C.byref(cols), C.byref(nnz), C.byref(a), C.byref(b), C.byref(v))

    def take(ptr, n, dt):
        out = np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].astype(dt, copy=True)
        ref.ref_free(C.cast(ptr, _p))
        return out
    return rows.value, cols.value, take(a, rows.value + 1, np.int32), take(b, nnz.value, np.int32), take(v, nnz.value, np.float32)


def cpu_spmv(row_ptr, col_idx, vals, x, y0, alpha, beta, rp_time=1):
    """cpu/src/main.cpp:11-23; returns the updated y (input not modified)."""
    rp, ci, va, xx = _c(row_ptr, np.int32), _c(col_idx, np.int32), _c(vals, np.float32), _c(x, np.float32)
    y = np.array(y0, dtype=np.float32, copy=True)
    lib.orc_cpu_spmv(rp.size - 1, _ptr(rp), _ptr(ci), _ptr(va), _ptr(xx), _ptr(y), alpha, beta, rp_time)
    return y


def cpu_sequential(coo_r, coo_c, coo_v, rows, x, cin, alpha, beta):
    """common/src/spmv-helper.cpp:812-833 (sparse branch)."""
    r, c, v = _c(coo_r, np.int32), _c(coo_c, np.int32), _c(coo_v, np.float32)
    xx, ci = _c(x, np.float32), _c(cin, np.float32)
    out = np.zeros(rows, dtype=np.float32)
    lib.orc_cpu_sequential(r.size, _ptr(r), _ptr(c), _ptr(v), rows, _ptr(xx), _ptr(ci), alpha, beta, _ptr(out))
    return out


def naive_gemv(A, x, y0, alpha, beta, rp_time=1):
    """cpu/src/main.cpp:53-71."""
    A = _c(A, np.float32)
    rows, cols = A.shape
    xx = _c(x, np.float32)
    y = np.array(y0, dtype=np.float32, copy=True)
    lib.orc_naive_gemv(rows, cols, _ptr(A), _ptr(xx), _ptr(y), alpha, beta, rp_time)
    return y


def precision_loss(a, b):
    """cpu/src/main.cpp:99-132 -> (precision_loss, max_rel_err, argmax)."""
    a, b = _c(a, np.float32), _c(b, np.float32)
    mre, mi = C.c_double(), C.c_int()
    pl = lib.orc_precision_loss(a.size, _ptr(a), _ptr(b), C.byref(mre), C.byref(mi))
    return pl, mre.value, mi.value


def spmv_f64(row_ptr, col_idx, vals, x, yin, alpha, beta):
    """fp64-accumulated truth and per-row magnitude |alpha| sum|a x| + |beta y| (not in the reference)."""
    rp, ci, va = _c(row_ptr, np.int32), _c(col_idx, np.int32), _c(vals, np.float32)
    xx, yy = _c(x, np.float32), _c(yin, np.float32)
    y = np.empty(rp.size - 1, dtype=np.float64)
    mag = np.empty(rp.size - 1, dtype=np.float64)
    lib.orc_spmv_f64(rp.size - 1, _ptr(rp), _ptr(ci), _ptr(va), _ptr(xx), _ptr(yy), alpha, beta, _ptr(y), _ptr(mag))
    return y, mag


def omp_spmv_timed(row_ptr, col_idx, vals, x, yin, alpha, beta, reps):
    """Timed OpenMP CSR SpMV ("port" CPU baseline) -> (seconds per rep, threads, y)."""
    rp, ci, va = _c(row_ptr, np.int32), _c(col_idx, np.int32), _c(vals, np.float32)
    xx, yy = _c(x, np.float32), _c(yin, np.float32)
    y = np.empty(rp.size - 1, dtype=np.float32)
    nt = C.c_int()
    t = lib.orc_omp_spmv_timed(rp.size - 1, _ptr(rp), _ptr(ci), _ptr(va), _ptr(xx), _ptr(yy), _ptr(y), alpha, beta, reps, C.byref(nt))
    return t, nt.value, y


def mkl_available() -> bool:
    return bool(lib.orc_mkl_available())


def mkl_spmv(row_ptr, col_idx, vals, cols, x, y0, alpha, beta, reps=1, threads=0):
    """mkl_sparse_s_mv as the reference calls it (cpu/src/main.cpp:26-49) -> (seconds per rep, threads, y) or None."""
    rp, ci, va, xx = _c(row_ptr, np.int32), _c(col_idx, np.int32), _c(vals, np.float32), _c(x, np.float32)
    y = np.array(y0, dtype=np.float32, copy=True)
    nt = C.c_int()
    t = lib.orc_mkl_spmv(rp.size - 1, cols, _ptr(rp), _ptr(ci), _ptr(va), _ptr(xx), _ptr(y), alpha, beta, reps, threads, C.byref(nt))
    if t < 0:
        return None
    return t, nt.value, y


def print_error_stats(cpu_ref, fpga_out):
    """The text HiSpmvHandle::printErrorStats would print (spmv-helper.cpp:835-895), or None where it throws / aborts."""
    a, b = _c(cpu_ref, np.float32), _c(fpga_out, np.float32)
    buf = C.create_string_buffer(1 << 16)
    n = lib.orc_print_error_stats(a.size, _ptr(a), b.size, _ptr(b), buf, len(buf))
    return None if n < 0 else buf.value.decode()


def cpu_bench_spmv(row_ptr, col_idx, vals, cols, mode, threads, budget_s, max_reps=200):
    """Baseline timing with first-touched copies (hispmv_oracle.cpp section 5) -> (seconds per rep, reps) or None."""
    rp, ci, va = _c(row_ptr, np.int32), _c(col_idx, np.int32), _c(vals, np.float32)
    reps = C.c_int()
    t = lib.orc_cpu_bench_spmv(rp.size - 1, int(cols), _ptr(rp), _ptr(ci), _ptr(va), int(mode), int(threads), float(budget_s), int(max_reps), C.byref(reps))
    return None if t < 0 else (t, reps.value)


def cpu_bench_gemv(rows, cols, mode, threads, budget_s, max_reps=10000):
    reps = C.c_int()
    t = lib.orc_cpu_bench_gemv(int(rows), int(cols), int(mode), int(threads), float(budget_s), int(max_reps), C.byref(reps))
    return None if t < 0 else (t, reps.value)


class RefPack:
    """Reference packed-stream format + dataflow emulator (refpack.inc)."""
    INFO = ("run_length", "rows_per_pe", "b_len", "padded_rows", "padded_cols", "tile_rows", "tile_cols",
            "row_tiles", "col_tiles", "output_length", "total_cycles", "words_per_channel", "num_pes", "dep_dist")

    def __init__(self, num_ch_A, num_ch_B, num_ch_C, urams_per_pe, fp_acc_latency, dense_overlay, pre_accumulator, row_dist_net):
        self.num_ch_A = num_ch_A
        self._h = lib.refpack_create(num_ch_A, num_ch_B, num_ch_C, urams_per_pe, fp_acc_latency,
                                     int(dense_overlay), int(pre_accumulator), int(row_dist_net))
        if not self._h:
            raise ValueError("bad hardware tuple")
        self.rows = self.cols = 0

    def __del__(self):
        if getattr(self, "_h", None):
            lib.refpack_free(self._h)
            self._h = None

    def prepare_sparse(self, rows, cols, r, c, v):
        r, c, v = _c(r, np.int32), _c(c, np.int32), _c(v, np.float32)
        self.rows, self.cols = rows, cols
        lib.refpack_prepare_sparse(self._h, rows, cols, r.size, _ptr(r), _ptr(c), _ptr(v))
        return self

    def prepare_dense(self, A):
        A = _c(A, np.float32)
        self.rows, self.cols = A.shape
        rc = lib.refpack_prepare_dense(self._h, self.rows, self.cols, _ptr(A))
        if rc != 0:
            raise AssertionError("dense mode unsupported for this tuple/shape")
        return self

    def info(self) -> dict:
        buf = np.zeros(len(self.INFO), dtype=np.int64)
        lib.refpack_info(self._h, _ptr(buf))
        return dict(zip(self.INFO, (int(t) for t in buf)))

    def channel(self, ch) -> np.ndarray:
        n = self.info()["words_per_channel"]
        return np.ctypeslib.as_array(lib.refpack_channel(self._h, ch), shape=(max(n, 1),))[:n].copy()

    def hash(self) -> int:
        return int(lib.refpack_hash(self._h))

    def count_bit(self, bit) -> int:
        return int(lib.refpack_count_bit(self._h, bit))

    def tile_size(self, i, j) -> int:
        return int(lib.refpack_tile_size(self._h, i, j))

    def shared_rows(self, i, j) -> np.ndarray:
        n = lib.refpack_shared_rows(self._h, i, j, None, 0)
        out = np.zeros(max(n, 1), dtype=np.int32)
        lib.refpack_shared_rows(self._h, i, j, _ptr(out), n)
        return out[:n]

    def emulate(self, x, cin, alpha, beta) -> np.ndarray:
        xx, ci = _c(x, np.float32), _c(cin, np.float32)
        y = np.zeros(self.rows, dtype=np.float32)
        lib.refpack_emulate(self._h, _ptr(xx), _ptr(ci), alpha, beta, _ptr(y))
        return y


def emu_spmv(words, hdr, fix, x, bias, alpha, beta, rows, mode=0):
    """CPU model of the product's slice kernel on the product's own stream (slice_emu.inc).
    mode 0 = slice kernel + fix-up kernels (the default product path); 1 = single launch with carry look-back."""
    words = _c(words, np.uint64)
    hdr = _c(hdr, np.int32).reshape(-1, 4)
    fix = _c(fix, np.int32).reshape(-1, 4)
    xx, bb = _c(x, np.float32), _c(bias, np.float32)
    y = np.zeros(rows, dtype=np.float32)
    lib.emu_spmv(_ptr(words), _ptr(hdr), _ptr(fix), hdr.shape[0], fix.shape[0], _ptr(xx), _ptr(bb), alpha, beta, _ptr(y), rows, mode)
    return y


def emu_tts(tts, x, bias, alpha, beta, rows):
    """CPU model of the transposed-tile-stream kernel on the product's own packed arrays (hispmv_amd.prep: Prepared.tts).
    A list of dicts = the column parts of the tall geometry: part 0 gives alpha*A_0*x + beta*bias, part t > 0 the partial
    vector alpha*A_t*x, added in part order (spmv_merge_multi_kernel: y = (y + part_1) + ...)."""
    if isinstance(tts, (list, tuple)):
        y = emu_tts(tts[0], x, bias, alpha, beta, rows)
        for part in tts[1:]:
            y = y + emu_tts(part, x, bias, alpha, 0.0, rows)
        return y
    w, cb, fl, ci = _c(tts["words"], np.uint32), _c(tts["col_base"], np.int32), _c(tts["flags"], np.uint16), _c(tts["chunk_info"], np.int32)
    ti, bl = _c(tts["tiles"], np.int32), _c(tts["blocks"], np.int32)
    xx, bb = _c(x, np.float32), _c(bias, np.float32)
    y = np.zeros(rows, dtype=np.float32)
    fx = _c(tts.get("fix", np.zeros((0, 4), np.int32)), np.int32)
    fh = tts.get("flags_hi")
    fh = _c(fh, np.uint16) if (fh is not None and np.size(fh)) else None        # gap-coded row ends (TtsGeometry::gap_rows)
    lib.emu_tts(_ptr(w), _ptr(cb), _ptr(fl), _ptr(ci), _ptr(ti), _ptr(bl), tts["n_tiles"], _ptr(fx), fx.shape[0], int(tts.get("n_carry", 0)),
                _ptr(xx), _ptr(bb), alpha, beta, _ptr(y), int(bool(tts.get("zero_fill", False))), _ptr(fh) if fh is not None else None)
    return y


def emu_gemv(W, x, bias, alpha, beta):
    W = _c(W, np.float32)
    xx, bb = _c(x, np.float32), _c(bias, np.float32)
    y = np.zeros(W.shape[0], dtype=np.float32)
    lib.emu_gemv(_ptr(W), W.shape[0], W.shape[1], _ptr(xx), _ptr(bb), alpha, beta, _ptr(y))
    return y
