"""Workload generators: seeded synthetic stand-ins for the reference's SuiteSparse benchmark set
(get_tb_matrices.py:57-78 downloads 20 matrices; there is no network here and none are checked in
-- SURVEY.md section 4) plus the power-law stress matrices of BASELINE.json configs[2].

Every stand-in keeps the real matrix's rows and (post-loader) nnz from BASELINE.md section 2 and is
drawn from one of two families:
  banded    FEM-like: row lengths ~ Poisson(nnz/rows), columns spread (jittered strata) over +-w of
            the diagonal, no forced runs of consecutive columns (pessimistic for x locality)
  scattered columns spread over the whole width; `powerlaw` adds bounded power-law row lengths
Real .mtx files, if the user drops them under matrices/<name>/<name>.mtx, are used instead.
"""
from __future__ import annotations

import zlib
from pathlib import Path

import numpy as np

# name, rows(=cols), nnz after the reference loader, family, half-bandwidth (banded only)
SUITESPARSE_SET = [
    ("PFlow_742", 742793, 37138400, "banded", 20000),
    ("soc-Pokec", 1632803, 30622600, "powerlaw", 0),
    ("mouse_gene", 45101, 28967300, "scattered", 0),
    ("TSOPF_RS_b2383", 38120, 16171200, "banded", 2400),
    ("Si41Ge41H72", 185639, 15011300, "banded", 30000),
    ("crankseg_2", 63838, 14148850, "banded", 6000),
    ("nd6k", 18000, 6897300, "banded", 3000),
    ("thread", 29736, 4444880, "banded", 3000),
    ("ASIC_680k", 682862, 2639000, "scattered", 0),
    ("nxp1", 414604, 2655880, "banded", 50000),
    ("analytics", 303813, 2006130, "scattered", 0),
    ("boyd2", 466316, 1500400, "banded", 30000),
    ("language", 399130, 1189850, "powerlaw", 0),
    ("crystk03", 24696, 1751180, "banded", 1500),
    ("trans5", 116835, 749800, "banded", 20000),
    ("ford2", 100196, 544690, "banded", 5000),
    ("lowThrust_7", 17378, 211560, "banded", 1000),
    ("c-52", 23948, 202710, "banded", 3000),
    ("hangGlider_3", 10260, 92700, "banded", 500),
    ("poli_large", 15575, 33030, "banded", 2000),
]


def algorithmic_bytes(rows: int, cols: int, nnz: int, beta_nonzero: bool = True) -> int:
    """8*nnz (fp32 value + int32 column) + 4*(rows+1) row pointers + 4*cols (x once)
    + 4*rows (y written) + 4*rows (bias read, when beta != 0); independent of our packing."""
    return 8 * nnz + 4 * (rows + 1) + 4 * cols + 4 * rows + (4 * rows if beta_nonzero else 0)


def flops(rows: int, nnz: int) -> int:
    """The reference's convention everywhere: 2*(nnz + rows) (spmv-host.cpp:100,185; cpu/src/main.cpp:187)."""
    return 2 * (nnz + rows)


def _row_lengths(rng, rows: int, nnz: int, powerlaw: bool) -> np.ndarray:
    """Row lengths with an exact total: Poisson around the mean (or around a bounded power law,
    (k+100)^-0.8 over a random row order: max row ~ 3e-4 of nnz, like soc-Pokec's max out-degree),
    then +-1 corrections on random rows."""
    if powerlaw:
        w = 1.0 / (rng.permutation(rows) + 100.0) ** 0.8
        lam = w * (nnz / w.sum())
    else:
        lam = np.full(rows, nnz / rows)
    lens = rng.poisson(lam).astype(np.int64)
    diff = int(nnz - lens.sum())
    while diff != 0:
        if diff > 0:
            idx = rng.integers(0, rows, size=diff)
            np.add.at(lens, idx, 1)
        else:
            cand = np.nonzero(lens > 0)[0]
            idx = rng.choice(cand, size=min(-diff, cand.size), replace=False)
            lens[idx] -= 1
        diff = int(nnz - lens.sum())
    return lens


def synth_csr(rows: int, cols: int, nnz: int, family: str, bandwidth: int = 0, seed: int = 0):
    """-> (row_ptr int32 [rows+1], col_idx int32 ascending per row, values float32 in (-1,1) \\ {0}).
    Columns of a row of length L: one uniform draw in each of L equal strata of the row's column
    range (the band around the diagonal, or the whole width) -- ascending by construction, O(nnz)."""
    rng = np.random.default_rng(seed)
    lens = _row_lengths(rng, rows, nnz, family == "powerlaw")
    row_ptr = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(lens, out=row_ptr[1:])
    row_of = np.repeat(np.arange(rows, dtype=np.int32), lens)
    k = np.arange(nnz, dtype=np.int64) - row_ptr[:-1][row_of]          # index inside the row
    if family == "banded":
        w = int(bandwidth) if bandwidth > 0 else max(1000, cols // 16)
        centre = (np.arange(rows, dtype=np.int64) * cols) // rows
        lo = np.maximum(centre - w, 0)
        width = np.minimum(centre + w + 1, cols) - lo
    else:
        lo = np.zeros(rows, dtype=np.int64)
        width = np.full(rows, cols, dtype=np.int64)
    stratum = (width / np.maximum(lens, 1))[row_of]                     # float64 stratum width per element
    col = lo[row_of] + np.floor((k + rng.random(nnz)) * stratum).astype(np.int64)
    col = np.minimum(col, (lo + width - 1)[row_of]).astype(np.int32)
    val = rng.random(nnz, dtype=np.float32) * np.float32(2.0) - np.float32(1.0)
    val[val == 0] = np.float32(0.5)
    assert row_ptr[-1] == nnz < 2**31
    return row_ptr.astype(np.int32), col, val


def synth_banded(rows, cols, nnz, bandwidth, seed=0):
    return synth_csr(rows, cols, nnz, "banded", bandwidth, seed)


def suitesparse_standin(name: str):
    """-> (rows, cols, row_ptr, col_idx, values, source) for one matrix of SUITESPARSE_SET."""
    for n, rows, nnz, fam, bw in SUITESPARSE_SET:
        if n == name:
            seed = zlib.crc32(name.encode())
            rp, ci, va = synth_csr(rows, rows, nnz, fam, bw, seed)
            return rows, rows, rp, ci, va, f"synthetic:{fam}"
    raise KeyError(name)


def real_matrix_path(name: str, root: Path | None = None) -> Path | None:
    root = root or Path(__file__).resolve().parents[1] / "matrices"
    p = root / name / f"{name}.mtx"
    return p if p.exists() else None


def rmat_coo(scale: int, edge_factor: int = 16, a=0.57, b=0.19, c=0.19, seed: int = 42):
    """R-MAT edge list (duplicates kept), BASELINE.json configs[2] / SURVEY.md 8d C3."""
    rng = np.random.default_rng(seed)
    n = 1 << scale
    m = edge_factor * n
    r = np.zeros(m, dtype=np.int64)
    cc = np.zeros(m, dtype=np.int64)
    for _ in range(scale):
        q = rng.random(m)
        down = q >= a + b                    # quadrants c,d -> lower half
        right = ((q >= a) & (q < a + b)) | (q >= a + b + c)
        r = (r << 1) | down
        cc = (cc << 1) | right
    v = rng.random(m, dtype=np.float32) + np.float32(0.001)
    return n, n, r.astype(np.int32), cc.astype(np.int32), v


def write_mtx(path, rows: int, cols: int, r, c, v, symmetry: str = "general", comment: str = "hispmv_amd") -> None:
    with open(path, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate real {symmetry}\n% {comment}\n{rows} {cols} {len(r)}\n")
        for i, j, x in zip(r, c, v):
            f.write(f"{int(i) + 1} {int(j) + 1} {float(x):.9g}\n")
