#!/usr/bin/env python3
"""Counterpart of the reference's benchmark driver common/src/spmv-host.cpp (:41-191) on MI355X.

    python examples/spmv_host.py <matrix.mtx> [--exec_ms 100] [--device 0]
    python examples/spmv_host.py <rows> <cols>          # dense overlay, A_ij from generateVector

Same inputs (alpha = 0.55, beta = -2.05, vectors (i+2)/(i+1): spmv-host.cpp:17-23,43-44), same FLOP
convention 2*(nnz+rows) (:100,:185) and the same labelled output lines, so the reference's log scraper
(builds/collect_data.py:8-21) parses these logs unchanged ("FPGA TIME"/"FPGA GFLOPS" carry the GPU numbers;
"Matrix A Length" is the number of wavefront slices, "Approx. Clock Cycles" slices x 8 steps).
The CPU check is the script's own fp32 CSR product (like the reference's cpuSequential, :98), followed by
the reference's relative-error histogram (spmv-helper.cpp:835-895).
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyhispmv  # noqa: E402
from hispmv_amd.report import print_error_stats  # noqa: E402

ALPHA, BETA = 0.55, -2.05


def generate_vector(n):
    i = np.arange(n, dtype=np.float32)
    return (np.float32(1.0) * (i + 2) / (i + 1)).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("matrix", nargs="+", help="<file.mtx>  or  <rows> <cols>")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--exec_ms", type=float, default=100.0)
    a = ap.parse_args()
    import torch
    fpga = pyhispmv.FpgaHandle("hispmv.xclbin", a.device, 24, 1, 1, 2, 5, True, False, True)
    fpga.set_arena_bytes(200 << 30)
    t0 = time.time()
    if len(a.matrix) == 1:
        idx = fpga.create_sparse_handle_from_mtx(a.matrix[0], 0)          # loadMtx semantics (spmv-helper.cpp:34-136)
        from hispmv_amd.prep import prep_from_mtx
        P = prep_from_mtx(a.matrix[0], 0)
        A = sp.csr_matrix((P.values, P.col_idx, P.row_ptr), shape=(P.rows, P.cols))
        rows, cols, nnz = P.rows, P.cols, P.nnz
        print(f"\nMatrix Properties:\n\tRows: {rows}\n\tCols: {cols}\n\tNNZ: {nnz}\n")
    else:
        rows, cols = int(a.matrix[0]), int(a.matrix[1])
        dense = generate_vector(rows * cols).reshape(rows, cols)           # spmv-host.cpp:76
        idx = fpga.create_dense_handle(dense.reshape(-1), rows, cols)
        A, nnz = dense, rows * cols
    assert idx >= 0, "matrix does not fit the arena"
    fpga.load_matrices()
    info = fpga.matrix_info(idx)
    print(f"Pre-processing Time: {info['prep_seconds']:.6f} secs (file parse + upload: {time.time() - t0 - info['prep_seconds']:.3f} s)")

    x, c_in = generate_vector(cols), generate_vector(rows)
    print("\nComputing on CPU... ")
    t = time.perf_counter()
    y_cpu = (np.float32(ALPHA) * (A @ x).astype(np.float32) + np.float32(BETA) * c_in).astype(np.float32)
    t_cpu = time.perf_counter() - t
    print(f"CPU TIME: {t_cpu * 1e3:.6f} ms")
    print(f"CPU GFLOPS: {2.0 * (nnz + rows) / t_cpu / 1e9:.6f}")
    print(f"Matrix A Length: {info['n_slices']}")
    print(f"Approx. Clock Cycles: {info['n_slices'] * 8}")

    dx, dc = torch.from_numpy(x).cuda(a.device), torch.from_numpy(c_in).cuda(a.device)
    dy = torch.zeros(rows, dtype=torch.float32, device=f"cuda:{a.device}")
    one = fpga.time_device(idx, dx.data_ptr(), dc.data_ptr(), dy.data_ptr(), ALPHA, BETA, 3)
    reps = int(max(1, min(1 << 15, a.exec_ms / max(one, 1e-4))))          # rp_time sized to exec_ms (:121-125)
    print(f"Using Repeat Time: {reps}")
    print("Using Num samples: 1")
    print("\nComputing on GPU...")
    ms = fpga.time_device(idx, dx.data_ptr(), dc.data_ptr(), dy.data_ptr(), ALPHA, BETA, reps)
    power = []
    try:                                      # FpgaPowerMonitor's role (spmv-host.cpp:113-141), when the driver exposes it
        for _ in range(5):
            fpga.time_device(idx, dx.data_ptr(), dc.data_ptr(), dy.data_ptr(), ALPHA, BETA, max(1, reps // 5))
            p_now = float(torch.cuda.power_draw(a.device))          # mW by the documentation; W from the ROCm backend here
            power.append(p_now / 1000.0 if p_now > 5000.0 else p_now)
    except Exception:
        power = []
    if power:
        for w in power:
            print(f"sample: {w:.3f}")
        print(f"Average Power: {sum(power) / len(power):.3f} Watts")
        print(f"Max Power: {max(power):.3f} Watts")
    print(f"Total Kernel Runtime: {ms * reps:.6f}ms ")
    print(f"FPGA TIME: {ms * 1e3:.4f}us ")
    print(f"FPGA GFLOPS: {2.0 * (nnz + rows) / (ms * 1e-3) / 1e9:.4f}")
    alg = (8 * nnz + 16 * rows + 4) if len(a.matrix) == 1 else (4 * nnz + 4 * cols + 8 * rows)
    print(f"Algorithmic HBM GB/s: {alg / (ms * 1e-3) / 1e9:.1f} ({100 * alg / (ms * 1e-3) / 8e12:.1f} % of 8 TB/s; launch plan "
          f"{info['block_threads']} threads, {info['group_slices']} slices/workgroup, {info['lds_bytes']} B LDS window, {info['col_tiles']} column tile(s))")
    print("\nComparing Results... ")
    print_error_stats(y_cpu, dy.cpu().numpy())
    fpga.close()


if __name__ == "__main__":
    main()
