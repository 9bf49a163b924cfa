#!/usr/bin/env python3
"""Host preprocessing time by phase for stand-ins of the bench set (no GPU needed):
   COO -> CSR -> slice stream (hispmv_prep_from_coo) and the launch planner (hispmv_prep_plan)."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from hispmv_amd import matrices as M
from hispmv_amd.prep import prep_from_coo
from hispmv_amd._lib import lib

for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["PFlow_742", "soc-Pokec", "mouse_gene", "ASIC_680k"]):
    rows, cols, rp, ci, va, _ = M.suitesparse_standin(name)
    r = np.repeat(np.arange(rows, dtype=np.int32), np.diff(rp))
    perm = np.random.default_rng(0).permutation(r.size)          # COO in random order, as a caller may hand it over
    r, c, v = r[perm], ci[perm], va[perm]
    t = time.time()
    P = prep_from_coo(r, c, v, rows, cols)
    t_prep = time.time() - t
    print(f"{name:12s} nnz {r.size:9d}  coo->csr->stream+plan(prep_from_coo) {t_prep:6.2f} s   slices {P.n_slices}")
