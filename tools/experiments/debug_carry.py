import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from hispmv_amd.prep import prep_from_coo
rng = np.random.default_rng(3)
rows, cols, nnz = 20000, 15000, 400000
r = rng.integers(0, rows, nnz); r[:100000] = 1234; r[r % 5 == 0] += 1
c = rng.integers(0, cols, nnz); v = rng.random(nnz, dtype=np.float32) - 0.5
x = rng.random(cols, dtype=np.float32); b = rng.random(rows, dtype=np.float32)
P = prep_from_coo(r, c, v, rows, cols)
eoff = np.concatenate([[0], np.cumsum(np.maximum(np.diff(P.row_ptr), 1))])
for carry, plan in (("fixup", "global"), ("fixup", "0"), ("fixup", "1"), ("fixup", "3")):
    os.environ["HISPMV_CARRY"] = carry; os.environ["HISPMV_PLAN"] = plan
    import pyhispmv
    h = pyhispmv.FpgaHandle("d.xclbin", 0, 24, 1, 1, 2, 5, True, False, True)
    idx = h.create_sparse_handle(r, c, v, rows, cols); h.load_matrices(); h.select_matrix(idx)
    info = h.matrix_info(idx)
    for it in range(1):
        y = np.full(rows, np.nan, np.float32); h.run_kernel(x, b, y, 0.85, -2.06)
        ye = oracle.emu_spmv(P.words, P.hdr, P.fix, x, b, 0.85, -2.06, rows, info["carry_lookback"])
        bad = np.nonzero(y.view(np.uint32) != ye.view(np.uint32))[0]
        y64, mag = oracle.spmv_f64(P.row_ptr.astype(np.int32), P.col_idx, P.values, x, b, 0.85, -2.06)
        if bad.size: print("   first bad", bad[0], y[bad[0]], ye[bad[0]], y64[bad[0]], "rowlen", int(np.diff(P.row_ptr)[bad[0]]))
        print(plan, carry, "lookback" if info["carry_lookback"] else "fixup", info["block_threads"], info["group_slices"], info["lds_bytes"], "iter", it, "mismatches", bad.size, bad[:8], [(int(eoff[i] // 1024), int((eoff[i + 1] - 1) // 1024)) for i in bad[:8]])
    h.close()
