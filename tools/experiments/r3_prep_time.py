#!/usr/bin/env python3
"""End-to-end time of create_sparse_handle + load_matrices for soc-Pokec's shape handed over as 30.6 M UNSORTED COO entries
(verdict item 9: the reference's published preprocessing time for this matrix is 18.46 s, builds/U280_metrics.csv:3)."""
import sys, time
import numpy as np
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[2]))
from hispmv_amd import matrices as M
import pyhispmv

name = sys.argv[1] if len(sys.argv) > 1 else "soc-Pokec"
rows, cols, rp, ci, va, _ = M.suitesparse_standin(name)
r = np.repeat(np.arange(rows, dtype=np.int32), np.diff(rp))
perm = np.random.default_rng(0).permutation(r.size)
r, c, v = r[perm], ci[perm], va[perm]
h = pyhispmv.FpgaHandle("x.xclbin", 0, 24, 1, 1, 2, 5, True, False, True)
h.set_arena_bytes(64 << 30)
for rep in range(3):
    t0 = time.perf_counter()
    idx = h.create_sparse_handle(r, c, v, rows, cols)
    t1 = time.perf_counter()
    h.load_matrices()
    t2 = time.perf_counter()
    info = h.matrix_info(idx)
    print(f"{name} rep {rep}: create_sparse_handle {t1 - t0:.3f} s (library prep_seconds {info['prep_seconds']:.3f}), load_matrices {t2 - t1:.3f} s, format {info['format']}", flush=True)
h.close()
