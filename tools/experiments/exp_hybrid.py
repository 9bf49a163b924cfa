#!/usr/bin/env python3
"""Experiment: matrices whose groups do not quite fit an LDS window (outlier couplings, x a little larger than
LDS).  Run under different HISPMV_PLAN settings; prints plan and us per SpMV (back-to-back launches)."""
import os, sys
import numpy as np
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
import torch
import pyhispmv

rng = np.random.default_rng(5)
cases = {}
rows = 400000
r = np.repeat(np.arange(rows, dtype=np.int32), 24)
for far_pct in (0, 2, 8, 20):
    c = (r + rng.integers(-300, 300, r.size)) % rows
    far = rng.random(r.size) < far_pct / 100
    cases[f"banded300 x24, {far_pct}% far"] = (rows, rows, r, np.where(far, rng.integers(0, rows, r.size), c).astype(np.int32))
rows2, cols2 = 45000, 45000
r2 = np.repeat(np.arange(rows2, dtype=np.int32), 640)
cases["45k cols uniform x640"] = (rows2, cols2, r2, rng.integers(0, cols2, r2.size).astype(np.int32))
h = pyhispmv.FpgaHandle("none", 0, 24, 1, 1, 2, 5, True, False, True)
h.set_arena_bytes(64 << 30)
ids = {}
for name, (rows, cols, r, c) in cases.items():
    ids[name] = h.create_sparse_handle(r, c, (rng.random(r.size, dtype=np.float32) - 0.5), rows, cols)
h.load_matrices()
for name, (rows, cols, r, c) in cases.items():
    x = torch.rand(cols, device="cuda"); b = torch.rand(rows, device="cuda"); y = torch.empty(rows, device="cuda")
    i = ids[name]
    h.time_device(i, x.data_ptr(), b.data_ptr(), y.data_ptr(), 0.85, -2.06, 5)
    ms = min(h.time_device(i, x.data_ptr(), b.data_ptr(), y.data_ptr(), 0.85, -2.06, 20) for _ in range(3))
    info = h.matrix_info(i)
    bytes_ = 8 * r.size + 16 * rows
    print(f"{os.environ.get('HISPMV_PLAN', 'auto'):7s} {name:28s} {ms * 1e3:8.1f} us {bytes_ / ms / 1e6:8.0f} GB/s  {info['block_threads']}t/{info['group_slices']}s/{info['lds_bytes'] >> 10}KiB/{info['col_tiles']}ct")
