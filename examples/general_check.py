#!/usr/bin/env python3
"""Counterpart of the reference's apps/general_test.py (same call sequence and sizes, seeded):
dense 50000x10000 GeMV + 1 M-nnz random COO SpMV through pyhispmv.FpgaHandle, checked with the script's
own criterion np.allclose(rtol=1e-3) (apps/general_test.py:106,113) and with the 1e-5 backward-error gate.

    python examples/general_check.py [--rows 50000 --cols 10000 --nnz 1000000]
"""
import argparse
import os
import sys
import time

import numpy as np
from scipy.sparse import coo_matrix

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyhispmv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=50000)
    ap.add_argument("--cols", type=int, default=10000)
    ap.add_argument("--nnz", type=int, default=1000000)
    a = ap.parse_args()
    np.random.seed(0)                                          # the reference script is unseeded
    fpga = pyhispmv.FpgaHandle("builds/Dense-HI-SpMV-24-1-1/SpMV.xclbin", 0, 24, 1, 1, 2, 5, True, False, True)
    rows, cols = a.rows, a.cols
    dense_values = np.random.rand(rows, cols).astype(np.float32)
    x = np.random.rand(cols).astype(np.float32)
    bias = np.random.rand(rows).astype(np.float32)
    y_dense = np.zeros(rows, dtype=np.float32)
    t = time.time(); y_dense_expected = np.dot(dense_values, x) + bias
    print(f"Dense matrix computation time (NumPy): {time.time() - t:.4f} seconds")
    coo_rows = np.random.randint(0, rows, size=a.nnz, dtype=np.int32)
    coo_cols = np.random.randint(0, cols, size=a.nnz, dtype=np.int32)
    coo_values = np.random.rand(a.nnz).astype(np.float32)
    y_sparse = np.zeros(rows, dtype=np.float32)
    sparse_matrix = coo_matrix((coo_values, (coo_rows, coo_cols)), shape=(rows, cols))
    t = time.time(); y_sparse_expected = sparse_matrix.dot(x) + bias
    print(f"Sparse matrix computation time (NumPy): {time.time() - t:.4f} seconds")

    t = time.time(); dense_idx = fpga.create_dense_handle(dense_values.flatten(), rows, cols)
    sparse_idx = fpga.create_sparse_handle(coo_rows, coo_cols, coo_values, rows, cols)
    fpga.load_matrices()
    print(f"create + load: {time.time() - t:.3f} s (handles {dense_idx}, {sparse_idx})")

    t = time.time(); fpga.select_matrix(dense_idx); fpga.run_kernel(x, bias, y_dense, 1.0, 1.0)
    print(f"GPU execution time for dense matrix: {time.time() - t:.4f} seconds (kernel {fpga.last_kernel_ms() * 1e3:.1f} us, "
          f"{4.0 * rows * cols / fpga.last_kernel_ms() / 1e6:.0f} GB/s)")
    t = time.time(); fpga.select_matrix(sparse_idx); fpga.run_kernel(x, bias, y_sparse, 1.0, 1.0)
    print(f"GPU execution time for sparse matrix: {time.time() - t:.4f} seconds (kernel {fpga.last_kernel_ms() * 1e3:.1f} us)")

    ok = True
    for name, y, ref in (("Dense", y_dense, y_dense_expected), ("Sparse", y_sparse, y_sparse_expected)):
        print(f"Maximum Absolute Error for {name} Matrix: {np.max(np.abs(y - ref)):.6f}")
        print(f"Maximum Relative Error for {name} Matrix: {np.max(np.abs(y - ref) / np.abs(ref)):.6f}")
        good = np.allclose(y, ref, rtol=1e-3)
        print(f"{name} matrix result is {'correct' if good else 'incorrect'}!")
        ok &= bool(good)
    d64 = dense_values.astype(np.float64) @ x.astype(np.float64) + bias
    mag = np.abs(dense_values.astype(np.float64)) @ np.abs(x.astype(np.float64)) + np.abs(bias)
    e_d = float(np.max(np.abs(y_dense - d64) / mag))
    s64 = sparse_matrix.astype(np.float64).dot(x.astype(np.float64)) + bias
    smag = abs(sparse_matrix).astype(np.float64).dot(np.abs(x.astype(np.float64))) + np.abs(bias)
    e_s = float(np.max(np.abs(y_sparse - s64) / smag))
    print(f"backward error vs fp64: dense {e_d:.2e}, sparse {e_s:.2e} (gate 1e-5)")
    fpga.close()
    sys.exit(0 if ok and e_d < 1e-5 and e_s < 1e-5 else 1)


if __name__ == "__main__":
    main()
