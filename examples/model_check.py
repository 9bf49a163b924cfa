#!/usr/bin/env python3
"""Counterpart of the reference's apps/model_test.py + apps/fpga_layer_manager.py + apps/model.py:
three-layer FC model 4096 -> 8192 (dense) -> 8192 (sparse, density 0.1) -> 1024 (sparse, density 0.25),
batch 1, every layer called rp_time=100 times through `FpgaHandle.linear` with ReLU between layers
(model.py:68-80), weights seeded with torch.manual_seed(0).  The reference compares against torch /
sparse_dot_mkl and prints error histograms; here the comparison is against an fp64 evaluation.

    python examples/model_check.py [--batch_size 1 --rp_time 100]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyhispmv import FpgaHandle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch_size", type=int, default=1)
    ap.add_argument("--input_size", type=int, default=4096)
    ap.add_argument("--hidden_size_1", type=int, default=8192)
    ap.add_argument("--hidden_size_2", type=int, default=8192)
    ap.add_argument("--output_size", type=int, default=1024)
    ap.add_argument("--density1", type=float, default=0.1)
    ap.add_argument("--density2", type=float, default=0.25)
    ap.add_argument("--rp_time", type=int, default=100)
    a = ap.parse_args()
    torch.manual_seed(0)
    fpga = FpgaHandle("builds/Dense-HI-SpMV-24-1-1/SpMV.xclbin", 0, 24, 1, 1, 2, 5, True, False, True)

    def sparse_weight(out_f, in_f, density):                       # model.py:21-31
        w = torch.randn((out_f, in_f))
        return (w * (torch.rand_like(w) < density)).to_sparse()

    lin = torch.nn.Linear(a.input_size, a.hidden_size_1)           # model.py:60
    layers = [("dense", lin.weight.detach().numpy(), lin.bias.detach().numpy())]
    for out_f, in_f, dens in ((a.hidden_size_2, a.hidden_size_1, a.density1), (a.output_size, a.hidden_size_2, a.density2)):
        sw = sparse_weight(out_f, in_f, dens)
        layers.append(("sparse", sw, np.zeros(out_f, np.float32)))
    handles = []
    t = time.time()
    for kind, w, b in layers:                                      # fpga_layer_manager.py:15-52
        if kind == "sparse":
            idx = fpga.create_sparse_handle(w._indices()[0].numpy(), w._indices()[1].numpy(), w._values().numpy(), *w.shape)
        else:
            density = np.count_nonzero(w) / w.size
            idx = fpga.create_dense_handle(w.flatten(), *w.shape) if density > 0.5 else None
        if idx == -1:
            raise RuntimeError("FPGA memory is full.")
        handles.append(idx)
    fpga.load_matrices()
    print(f"create + load of 3 layers: {time.time() - t:.2f} s")

    x = torch.randn((a.batch_size, a.input_size)).numpy()
    h = x
    h64 = x.astype(np.float64)
    worst = 0.0
    for (kind, w, b), idx in zip(layers, handles):
        t = time.time()
        for _ in range(a.rp_time):
            y = fpga.linear(idx, h.reshape(-1), b)
        dt = (time.time() - t) / a.rp_time
        y = y.reshape(a.batch_size, -1)
        wd = (w.to_dense().numpy() if kind == "sparse" else w).astype(np.float64)
        y64 = h.astype(np.float64) @ wd.T + b
        mag = np.abs(h.astype(np.float64)) @ np.abs(wd.T) + np.abs(b)
        err = float(np.max(np.abs(y - y64) / mag))
        worst = max(worst, err)
        print(f"layer {idx} ({kind} {wd.shape[0]}x{wd.shape[1]}): {dt * 1e6:.1f} us per linear() call incl. PCIe, "
              f"device {fpga.last_kernel_ms() * 1e3:.1f} us, backward error {err:.2e}")
        h = np.maximum(y, 0).astype(np.float32)
    fpga.close()
    print("model check", "passed" if worst < 1e-5 else "FAILED", f"(worst backward error {worst:.2e})")
    sys.exit(0 if worst < 1e-5 else 1)


if __name__ == "__main__":
    main()
