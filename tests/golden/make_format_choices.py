#!/usr/bin/env python3
"""Pins the loader's format / tiling decision (hispmv_amd/csrc/hispmv_choose.cpp, through the host-only entry
hispmv_prep_choose_format) for every matrix the benchmarks and parity tests run: the 20 shapes of the SuiteSparse set in
both stand-in families, the C3 power-law / adversarial matrices and the sparse C4 layers of apps/model_test.py, for a 256-CU
device -> tests/golden/format_choices.json.  Regenerate ON PURPOSE when the planner changes (the diff of the JSON is the
review of that change):  python tests/golden/make_format_choices.py"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

KEYS = ("format", "tile_kind", "parts", "tile_width", "tile_base", "l2_tiles", "threads", "group", "lds_floats", "n_slices", "n_elems", "n_split")


def cases():
    """Yields (name, rows, cols, row_ptr, col_idx, values), one matrix at a time (the set is ~170 M entries)."""
    from hispmv_amd import matrices as M
    for uniform in (False, True):
        for name, rows, nnz, fam, par in M.SUITESPARSE_SET:
            if uniform and fam != "fem":
                continue                      # the other 12 are the same matrix in both families
            r, c, rp, ci, va, _src = M.suitesparse_standin(name, uniform)
            yield f"{'uniform' if uniform else 'structured'}:{name}", r, c, rp, ci, va
    n, _, r, c, v = M.rmat_coo(20)
    rp, ci, va = M.coo_to_csr_sorted(r, c, v, n)
    yield "C3:rmat20", n, n, rp, ci, va
    rp, ci, va = M.zipf_csr(1632803, 1632803, 30622600, 1.2, 7)
    yield "C3:zipf1.2_pokec_shape", 1632803, 1632803, rp, ci, va
    rp, ci, va = M.heavy_rows_csr(400000, 400000, 8000000)
    yield "C3:1pct_rows_90pct_nnz", 400000, 400000, rp, ci, va
    rp, ci, va = M.full_row_plus_diagonal(1000000)
    yield "C3:full_row_plus_diag", 1000000, 1000000, rp, ci, va
    for idx, (kind, W, rows, cols, _bias) in enumerate(M.model_test_layers(0)):
        if kind != "dense":
            rp, ci, va = M.coo_to_csr_sorted(W[0], W[1], W[2], rows)
            yield f"C4:model_layer{idx}", rows, cols, rp, ci, va


def decide(rows, cols, rp, ci, va):
    from hispmv_amd.prep import choose_format_from_csr
    d = choose_format_from_csr(rp, ci, va, rows, cols, 256)
    return {k: d[k] for k in KEYS}


def main():
    out = {}
    for name, rows, cols, rp, ci, va in cases():
        t0 = time.time()
        out[name] = decide(rows, cols, rp, ci, va)
        print(f"{name:40s} {out[name]}  ({time.time() - t0:.1f} s)", flush=True)
    (Path(__file__).resolve().parent / "format_choices.json").write_text(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
