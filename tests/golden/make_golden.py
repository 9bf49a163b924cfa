#!/usr/bin/env python3
"""Generates the committed golden fixtures of tests/golden/.

Run IN THE BUILD CONTAINER (needs /root/reference for oracle/_ref):

    make -C oracle && python tests/golden/make_golden.py

What is produced, and where each array comes from (also written to provenance.json):
  <case>.mtx        seeded synthetic MatrixMarket inputs (this script; the reference ships no
                    matrices and SuiteSparse cannot be downloaded here -- SURVEY.md section 4).
  <case>.npz
     ref_row_ptr / ref_col_idx / ref_vals
                    output of the REFERENCE's own loader readMatrixCSC + convertCSCtoCSR
                    (cpu/src/helper_functions.cpp:148-241), compiled from /root/reference by
                    oracle/Makefile into oracle/_ref/libref_cpu.so.  This is the
                    "bit-exact on indices" target of BASELINE.json.
     y_mkl          mkl_sparse_s_mv on those arrays, called exactly as cpu/src/main.cpp:26-49
                    does (oracle.mkl_spmv, dlopen of the image's libmkl_rt), with the reference's
                    deterministic vectors x_j=(j+1)/(j+2), y_i=-2(i+1)/(i+2), alpha=0.85,
                    beta=-2.06 (main.cpp:147-148,173-178), reps=1.  Third-party arithmetic.
     y_cpu_spmv     the oracle's restatement of cpu_spmv (main.cpp:11-23) on the same inputs.
     coo_r / coo_c / coo_v
                    the oracle's restatement of HiSpmvHandle::loadMtx (spmv-helper.cpp:34-136);
                    a regression anchor only (common/ cannot be compiled here: needs tapa.h/xrt).
No reference source text is stored; only inputs and outputs.
"""
from __future__ import annotations

import hashlib
import json
import random
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402

OUT = Path(__file__).resolve().parent
ALPHA, BETA = 0.85, -2.06


def write_mtx(path: Path, kind: str, sym: str, rows: int, cols: int, entries, comment: str):
    with open(path, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate {kind} {sym}\n")
        f.write(f"% {comment}\n")
        f.write(f"{rows} {cols} {len(entries)}\n")
        for e in entries:
            if kind == "pattern":
                f.write(f"{e[0]} {e[1]}\n")
            else:
                f.write(f"{e[0]} {e[1]} {e[2]}\n")


def case_syn_1138():
    # KAT-1 of SURVEY.md Appendix C.2, verbatim procedure
    random.seed(1138)
    n = 1138
    ents = {}
    for i in range(n):
        ents[(i, i)] = round(random.uniform(0.5, 5.0), 6)
    while len(ents) < 2596:
        i = random.randrange(n)
        j = random.randrange(n)
        if i > j:
            ents[(i, j)] = round(random.uniform(-2.0, -0.01), 6)
    items = sorted(ents.items(), key=lambda kv: (kv[0][1], kv[0][0]))
    entries = [(i + 1, j + 1, v) for (i, j), v in items]
    return "real", "symmetric", n, n, entries, "synthetic stand-in for HB/1138_bus"


def case_gen_real():
    rng = random.Random(11)
    rows, cols = 300, 200
    seen = {}
    while len(seen) < 1500:
        seen[(rng.randrange(rows), rng.randrange(cols))] = round(rng.uniform(-3, 3), 5)
    entries = [(i + 1, j + 1, v) for (i, j), v in seen.items()]   # insertion order = unsorted
    # explicit zeros in three spellings: dropped by both loaders except -0.0 (cpu/ keeps it)
    free = [(i, j) for i in range(5) for j in range(5) if (i, j) not in seen][:3]
    for (i, j), z in zip(free, ("0", "0.0", "-0.0")):
        entries.insert(rng.randrange(len(entries)), (i + 1, j + 1, z))
    return "real", "general", rows, cols, entries, "general real, unsorted, explicit zeros"


def case_skew():
    rng = random.Random(12)
    n = 150
    seen = {}
    while len(seen) < 600:
        i, j = rng.randrange(n), rng.randrange(n)
        if i > j:
            seen[(i, j)] = round(rng.uniform(-2, 2), 5)
    entries = [(i + 1, j + 1, v) for (i, j), v in sorted(seen.items(), key=lambda kv: (kv[0][1], kv[0][0]))]
    return "real", "skew-symmetric", n, n, entries, "skew-symmetric: mirrored+negated by common/, not mirrored by cpu/"


def case_pattern_sym():
    rng = random.Random(13)
    n = 120
    seen = set()
    while len(seen) < 500:
        i, j = rng.randrange(n), rng.randrange(n)
        if i >= j:
            seen.add((i, j))
    entries = [(i + 1, j + 1) for (i, j) in sorted(seen, key=lambda t: (t[1], t[0]))]
    return "pattern", "symmetric", n, n, entries, "pattern symmetric"


def case_integer_wide():
    rng = random.Random(14)
    rows, cols = 80, 300
    seen = {}
    while len(seen) < 900:
        seen[(rng.randrange(rows), rng.randrange(cols))] = rng.randint(-9, 9) or 1
    entries = [(i + 1, j + 1, v) for (i, j), v in seen.items()]
    return "integer", "general", rows, cols, entries, "integer general, rows << cols"


def case_tall_empty_rows():
    rng = random.Random(15)
    rows, cols = 2000, 30
    seen = {}
    while len(seen) < 2500:
        i = rng.randrange(rows)
        if i % 3 == 1:          # every third row stays empty
            continue
        seen[(i, rng.randrange(cols))] = round(rng.uniform(0.1, 1.0), 5)
    entries = [(i + 1, j + 1, v) for (i, j), v in seen.items()]
    return "real", "general", rows, cols, entries, "rows >> cols with empty rows"


def case_dense_row():
    rng = random.Random(16)
    n = 400
    seen = {(i, i): round(rng.uniform(1, 2), 5) for i in range(n)}
    for j in range(n):
        seen[(37, j)] = round(rng.uniform(-1, 1), 5)
    entries = [(i + 1, j + 1, v) for (i, j), v in sorted(seen.items(), key=lambda kv: (kv[0][1], kv[0][0]))]
    return "real", "general", n, n, entries, "diagonal plus one full row"


def case_powerlaw():
    rng = random.Random(17)
    n = 3000
    seen = {}
    weights = [1.0 / (k + 1) ** 1.1 for k in range(n)]
    rows_pick = rng.choices(range(n), weights=weights, k=40000)
    for i in rows_pick:
        if len(seen) >= 12000:
            break
        seen[(i, rng.randrange(n))] = round(rng.uniform(-1, 1), 5) or 0.5
    entries = [(i + 1, j + 1, v) for (i, j), v in seen.items()]
    return "real", "general", n, n, entries, "Zipf row lengths x uniform columns (rows split across slices)"


CASES = {
    "syn_1138": case_syn_1138, "gen_real": case_gen_real, "skew": case_skew, "pattern_sym": case_pattern_sym,
    "integer_wide": case_integer_wide, "tall_empty_rows": case_tall_empty_rows, "dense_row": case_dense_row,
    "powerlaw": case_powerlaw,
}


def main():
    if not oracle.ref_available():
        sys.exit("oracle/_ref/libref_cpu.so missing: run `make -C oracle` in the container that has /root/reference")
    prov = {"generator": "tests/golden/make_golden.py", "mkl_available": oracle.mkl_available(), "cases": {}}
    for name, fn in CASES.items():
        kind, sym, rows, cols, entries, comment = fn()
        mtx = OUT / f"{name}.mtx"
        write_mtx(mtx, kind, sym, rows, cols, entries, comment)
        r_rows, r_cols, rp, ci, va = oracle.ref_read_mtx_csr(mtx)
        x = ((np.arange(r_cols, dtype=np.float32) + 1) / (np.arange(r_cols, dtype=np.float32) + 2)).astype(np.float32)
        y0 = (np.float32(-2.0) * (np.arange(r_rows, dtype=np.float32) + 1) / (np.arange(r_rows, dtype=np.float32) + 2)).astype(np.float32)
        y_cpu = oracle.cpu_spmv(rp, ci, va, x, y0, ALPHA, BETA, 1)
        arrays = dict(rows=np.int32(r_rows), cols=np.int32(r_cols), ref_row_ptr=rp, ref_col_idx=ci, ref_vals=va,
                      y_cpu_spmv=y_cpu)
        m = oracle.mkl_spmv(rp, ci, va, r_cols, x, y0, ALPHA, BETA, reps=1, threads=1)
        if m is not None:
            arrays["y_mkl"] = m[2]
        c_rows, c_cols, cr, cc, cv = oracle.load_mtx_common(mtx)
        arrays.update(coo_r=cr, coo_c=cc, coo_v=cv)
        np.savez_compressed(OUT / f"{name}.npz", **arrays)
        prov["cases"][name] = {
            "mtx_md5": hashlib.md5(mtx.read_bytes()).hexdigest(), "rows": r_rows, "cols": r_cols,
            "nnz_cpu_loader": int(ci.size), "nnz_common_loader": int(cr.size), "has_y_mkl": m is not None,
        }
        print(name, prov["cases"][name])
    (OUT / "provenance.json").write_text(json.dumps(prov, indent=1) + "\n")


if __name__ == "__main__":
    main()
