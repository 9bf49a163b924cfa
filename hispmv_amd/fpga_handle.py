"""``FpgaHandle`` -- host-side mirror of the reference's pybind11 class
(pyhispmv/include/fpga_handle.h:9-74, pyhispmv/src/fpga_handle.cpp, bindings
pyhispmv/src/pyhispmv_bindings.cpp:3-39) over the C ABI of libhispmv.so.

Same method names, keyword names, argument meaning and return values as the reference, so
apps/general_test.py and apps/model_test.py run against it unchanged.  Differences, all on
the error path: the reference prints and calls ``std::exit`` (fpga_handle.cpp:58-64,82-88,
267-270) or ``assert``s (:292); here the same conditions raise Python exceptions.
There is no CPU execution path: every method that computes needs the gfx950 device.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib


def _ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


def _as(a, dtype) -> np.ndarray:
    # pybind11's py::array_t<T> default flags (c_style | forcecast): other dtypes / layouts are
    # converted by copy (e.g. torch int64 COO indices from apps/fpga_layer_manager.py:29-33).
    return np.ascontiguousarray(a, dtype=dtype)


class FpgaHandle:
    """y = alpha * A @ x + beta * bias on one MI355X; A sparse (COO in) or dense (row-major in)."""

    def __init__(self, xclbin_path: str, device_id: int, num_ch_A: int, num_ch_B: int, num_ch_C: int,
                 urams_per_pe: int, fp_acc_latency: int, dense_overlay: bool, pre_accumulator: bool,
                 row_dist_net: bool):
        self._ctx = C.c_void_p()
        rc = lib.hispmv_create(C.byref(self._ctx), str(xclbin_path).encode(), int(device_id),
                               int(num_ch_A), int(num_ch_B), int(num_ch_C), int(urams_per_pe),
                               int(fp_acc_latency), int(bool(dense_overlay)), int(bool(pre_accumulator)),
                               int(bool(row_dist_net)))
        if rc != _lib.HISPMV_OK:
            msg = lib.hispmv_last_error(None).decode()
            self._ctx = C.c_void_p()
            if rc == _lib.HISPMV_EINVAL:
                raise ValueError(f"Error initializing device: {msg}")
            raise RuntimeError(f"Error initializing device: {msg}")
        self.num_ch_A, self.num_ch_B, self.num_ch_C = num_ch_A, num_ch_B, num_ch_C
        self.device_id = device_id
        self._selected = None

    # -- lifetime -------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_ctx", None):
            lib.hispmv_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self) -> str:
        return lib.hispmv_last_error(self._ctx).decode()

    def _check(self, rc: int) -> int:
        if rc >= 0:
            return rc
        msg = self._err()
        if rc == _lib.HISPMV_EINVAL:
            raise IndexError(msg) if "out of range" in msg else ValueError(msg)
        if rc == _lib.HISPMV_ESTATE:
            raise AssertionError(msg)
        if rc == _lib.HISPMV_ENOTDENSE:
            raise AssertionError(msg)
        if rc == _lib.HISPMV_EIO:
            raise OSError(msg)
        if rc == _lib.HISPMV_ENOMEM:
            raise MemoryError(msg)
        raise RuntimeError(msg)

    # -- reference API (bindings :15-38) --------------------------------------------------------
    def create_dense_handle(self, flattened_dense_values, rows: int, cols: int) -> int:
        """Creates a matrix handle for a dense matrix; returns its index, or -1 if it does not fit."""
        a = _as(flattened_dense_values, np.float32).reshape(-1)
        if a.size < int(rows) * int(cols):
            raise ValueError("flattened_dense_values shorter than rows*cols")
        rc = lib.hispmv_create_dense_handle(self._ctx, _ptr(a), int(rows), int(cols))
        return rc if rc == _lib.HISPMV_FULL else self._check(rc)

    def create_sparse_handle(self, coo_rows, coo_cols, coo_values, rows: int, cols: int) -> int:
        """Creates a matrix handle for a sparse matrix (COO); returns its index, or -1 if it does not fit."""
        r = _as(coo_rows, np.int32).reshape(-1)
        c = _as(coo_cols, np.int32).reshape(-1)
        v = _as(coo_values, np.float32).reshape(-1)
        if not (r.size == c.size == v.size):
            raise ValueError("coo_rows, coo_cols and coo_values must have the same length")
        rc = lib.hispmv_create_sparse_handle(self._ctx, _ptr(r), _ptr(c), _ptr(v), r.size, int(rows), int(cols))
        return rc if rc == _lib.HISPMV_FULL else self._check(rc)

    def load_matrices(self) -> None:
        """Loads matrices into HBM (call after all create_* and before any run)."""
        self._check(lib.hispmv_load_matrices(self._ctx))

    def select_matrix(self, matrix_idx: int) -> None:
        """Select a matrix by its index."""
        if matrix_idx < 0:
            raise IndexError("Matrix idx out of range")
        self._check(lib.hispmv_select_matrix(self._ctx, int(matrix_idx)))
        self._selected = int(matrix_idx)

    def run_kernel(self, x, bias, y, alpha: float, beta: float) -> None:
        """Runs y = alpha*A*x + beta*bias for the selected matrix; y is written in place."""
        if not (isinstance(y, np.ndarray) and y.dtype == np.float32 and y.flags.c_contiguous and y.flags.writeable):
            raise TypeError("y must be a writable C-contiguous float32 numpy array (it is written in place)")
        xa = _as(x, np.float32).reshape(-1)
        ba = _as(bias, np.float32).reshape(-1)
        if self._selected is not None:   # the reference reads out of bounds instead
            info = self.matrix_info(self._selected)
            if xa.size < info["cols"] or y.size < info["rows"] or (beta != 0.0 and ba.size < info["rows"]):
                raise ValueError("vector shorter than the selected matrix dimension")
        self._check(lib.hispmv_run_kernel(self._ctx, _ptr(xa), _ptr(ba), _ptr(y), float(alpha), float(beta)))

    def linear(self, matrix_idx: int, x, bias) -> np.ndarray:
        """Run y = A*x + bias for each of the len(x)//cols vectors in the flattened x; returns a new array."""
        info = self.matrix_info(matrix_idx)
        xa = _as(x, np.float32).reshape(-1)
        ba = _as(bias, np.float32).reshape(-1)
        if ba.size < info["rows"]:
            raise ValueError("bias shorter than the matrix row dimension")
        num_vecs = xa.size // info["cols"]
        out = np.empty(num_vecs * info["rows"], dtype=np.float32)
        self._check(lib.hispmv_linear(self._ctx, int(matrix_idx), _ptr(xa), xa.size, _ptr(ba), _ptr(out)))
        return out

    # -- additions (no reference counterpart) ----------------------------------------------------
    def create_sparse_handle_from_mtx(self, path: str, flavor: int = 0) -> int:
        """MatrixMarket file -> handle (HiSpmvHandle::prepareSparseMtxForFPGA(mtx_file), spmv-helper.cpp:642)."""
        rc = lib.hispmv_create_sparse_handle_from_mtx(self._ctx, str(path).encode(), int(flavor))
        return rc if rc == _lib.HISPMV_FULL else self._check(rc)

    def create_sparse_handle_from_csr(self, row_ptr, col_idx, values, rows: int, cols: int) -> int:
        rp = _as(row_ptr, np.int32).reshape(-1)
        ci = _as(col_idx, np.int32).reshape(-1)
        va = _as(values, np.float32).reshape(-1)
        if rp.size != rows + 1 or ci.size != va.size or (rp.size and rp[-1] != ci.size):
            raise ValueError("inconsistent CSR arrays")
        rc = lib.hispmv_create_sparse_handle_from_csr(self._ctx, _ptr(rp), _ptr(ci), _ptr(va), int(rows), int(cols))
        return rc if rc == _lib.HISPMV_FULL else self._check(rc)

    def set_arena_bytes(self, nbytes: int) -> None:
        self._check(lib.hispmv_set_arena_bytes(self._ctx, int(nbytes)))

    def arena_bytes_used(self) -> int:
        return int(lib.hispmv_arena_bytes_used(self._ctx))

    def num_matrices(self) -> int:
        return int(lib.hispmv_num_matrices(self._ctx))

    def matrix_info(self, matrix_idx: int) -> dict:
        info = _lib.MatrixInfo()
        rc = lib.hispmv_get_matrix_info(self._ctx, int(matrix_idx), C.byref(info))
        if rc != _lib.HISPMV_OK:
            raise IndexError("Matrix idx out of range")
        return {k: getattr(info, k) for k, _ in _lib.MatrixInfo._fields_}

    def spmv_device(self, matrix_idx: int, d_x: int, d_bias: int, d_y: int, alpha: float, beta: float,
                    stream: int = 0) -> None:
        """Asynchronous launch on device pointers (ints), e.g. torch tensors' ``data_ptr()``."""
        self._check(lib.hispmv_spmv_device(self._ctx, int(matrix_idx), C.c_void_p(d_x), C.c_void_p(d_bias),
                                           C.c_void_p(d_y), float(alpha), float(beta), C.c_void_p(stream)))

    def prepare_batch(self, matrix_idxs, d_xs, d_biases, d_ys):
        """Argument block for `spmv_device_batch` (host arrays of device pointers), built once and reused."""
        n = len(matrix_idxs)
        if not (len(d_xs) == len(d_ys) == n and (d_biases is None or len(d_biases) == n)):
            raise ValueError("prepare_batch: lists of different length")
        idx = (C.c_int32 * n)(*[int(i) for i in matrix_idxs])
        xs = (C.c_void_p * n)(*[int(p) for p in d_xs])
        bs = (C.c_void_p * n)(*[int(p) for p in d_biases]) if d_biases is not None else None
        ys = (C.c_void_p * n)(*[int(p) for p in d_ys])
        return (n, idx, xs, bs, ys)

    def spmv_device_batch(self, batch, alpha: float, beta: float, stream: int = 0) -> None:
        """n independent SpMVs in as few launches as possible (hispmv_spmv_device_batch): matrices with the same
        workgroup size share one grid.  `batch` comes from `prepare_batch`.  Asynchronous on `stream`."""
        n, idx, xs, bs, ys = batch
        self._check(lib.hispmv_spmv_device_batch(self._ctx, n, idx, xs, bs, ys, float(alpha), float(beta), C.c_void_p(stream)))

    def time_device(self, matrix_idx: int, d_x: int, d_bias: int, d_y: int, alpha: float, beta: float,
                    reps: int) -> float:
        ms = lib.hispmv_time_device(self._ctx, int(matrix_idx), C.c_void_p(d_x), C.c_void_p(d_bias),
                                    C.c_void_p(d_y), float(alpha), float(beta), int(reps))
        if ms < 0:
            raise RuntimeError(self._err() or "time_device failed")
        return float(ms)

    def boundary_pack(self, d_last: int, d_mask: int, d_send: int, n: int, stream: int = 0) -> None:
        """send[i] = mask[i] * *last[i] (hispmv_boundary_pack); `stream` 0 = the context's stream, as for spmv_device."""
        self._check(lib.hispmv_boundary_pack(self._ctx, C.c_void_p(d_last), C.c_void_p(d_mask), C.c_void_p(d_send), int(n), C.c_void_p(stream)))

    def boundary_apply(self, d_first: int, d_recv: int, d_weights: int, n: int, world: int, stream: int = 0) -> None:
        """*first[i] += sum_r recv[r*n + i] * weights[i*world + r] (hispmv_boundary_apply); `stream` 0 = the context's stream."""
        self._check(lib.hispmv_boundary_apply(self._ctx, C.c_void_p(d_first), C.c_void_p(d_recv), C.c_void_p(d_weights), int(n), int(world),
                                              C.c_void_p(stream)))

    def synchronize(self) -> None:
        self._check(lib.hispmv_synchronize(self._ctx))

    def last_kernel_ms(self) -> float:
        return float(lib.hispmv_last_kernel_ms(self._ctx))

    def batch_graph_stats(self) -> dict:
        """{"instantiations", "alpha_updates"} of the HIP-graph replay of spmv_device_batch (hispmv_batch_graph_stats)."""
        import ctypes as C
        out = (C.c_int64 * 2)()
        self._check(lib.hispmv_batch_graph_stats(self._ctx, out))
        return {"instantiations": int(out[0]), "alpha_updates": int(out[1])}

    def batch_call_info(self) -> dict:
        """How the last spmv_device_batch call was issued (hispmv_batch_call_info): launches, whether its slice groups and tiles
        ran as items of the step kernel's queue, the items, the HIP streams of the main launches."""
        import ctypes as C
        out = (C.c_int64 * 4)()
        self._check(lib.hispmv_batch_call_info(self._ctx, out))
        return {"launches": int(out[0]), "step_kernel": bool(out[1]), "items": int(out[2]), "streams": int(out[3])}
