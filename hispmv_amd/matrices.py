"""Workload generators: seeded synthetic stand-ins for the reference's SuiteSparse benchmark set
(get_tb_matrices.py:57-78 downloads 20 matrices; there is no network here and none are checked in
-- SURVEY.md section 4) plus the power-law stress matrices of BASELINE.json configs[2].

Every stand-in keeps the real matrix's rows and (post-loader) nnz from BASELINE.md section 2 and is
drawn from one of two families:
  fem       structured-mesh / FEM-like (matrices that come from meshes): `dofs` unknowns per node, each node
            coupled to a fixed stencil of neighbour offsets (clusters of consecutive nodes inside a band),
            dense dofs x dofs blocks, couplings dropped at random to hit the exact nnz
  banded    row lengths ~ Poisson(nnz/rows), columns spread (jittered strata) over +-w of the diagonal, no
            column reuse between rows (circuit / optimisation matrices; also the pessimistic variant of the
            FEM ones: bench.py --standin uniform)
  scattered columns spread over the whole width; `powerlaw` adds bounded power-law row lengths
Real .mtx files, if the user drops them under matrices/<name>/<name>.mtx, are used instead.
"""
from __future__ import annotations

import zlib
from pathlib import Path

import numpy as np

# Half-bandwidths used when the FEM-origin matrices are generated with the unstructured `banded` family
# instead (bench.py --standin uniform): the pessimistic variant, no column reuse between rows.
UNIFORM_BAND = {"PFlow_742": 20000, "TSOPF_RS_b2383": 2400, "Si41Ge41H72": 30000, "crankseg_2": 6000, "nd6k": 3000,
                "thread": 3000, "crystk03": 1500, "ford2": 5000}

# name, rows(=cols), nnz after the reference loader, family, parameter:
#   fem -> (dofs per node, run of consecutive neighbour nodes, half band in nodes); banded -> half bandwidth
SUITESPARSE_SET = [
    ("PFlow_742", 742793, 37138400, "fem", (1, 4, 30000)),
    ("soc-Pokec", 1632803, 30622600, "powerlaw", 0),
    ("mouse_gene", 45101, 28967300, "scattered", 0),
    ("TSOPF_RS_b2383", 38120, 16171200, "fem", (8, 2, 300)),
    ("Si41Ge41H72", 185639, 15011300, "fem", (1, 8, 40000)),
    ("crankseg_2", 63838, 14148850, "fem", (3, 2, 8000)),
    ("nd6k", 18000, 6897300, "fem", (3, 2, 2500)),
    ("thread", 29736, 4444880, "fem", (3, 2, 3000)),
    ("ASIC_680k", 682862, 2639000, "scattered", 0),
    ("nxp1", 414604, 2655880, "banded", 50000),
    ("analytics", 303813, 2006130, "scattered", 0),
    ("boyd2", 466316, 1500400, "banded", 30000),
    ("language", 399130, 1189850, "powerlaw", 0),
    ("crystk03", 24696, 1751180, "fem", (3, 3, 1500)),
    ("trans5", 116835, 749800, "banded", 20000),
    ("ford2", 100196, 544690, "fem", (1, 2, 8000)),
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


def synth_fem(rows: int, nnz: int, dofs: int, run: int, half_band_nodes: int, seed: int = 0):
    """FEM / structured-mesh-like square matrix: `dofs` unknowns per node; node i couples to the nodes
    i + o_k for a fixed set of offsets o_k (clusters of `run` consecutive nodes inside +-half_band_nodes,
    like the planes/lines/points of a mesh stencil), each coupling kept with probability p so that the mean
    row length is nnz/rows; a coupling is a dense dofs x dofs block (runs of `dofs` consecutive columns; the
    `dofs` rows of a node share their column set; consecutive nodes have the same pattern shifted by one).
    Exactly `nnz` entries (random surplus entries are dropped).  Columns ascending per row."""
    rng = np.random.default_rng(seed)
    d = int(dofs)
    n_nodes = -(-rows // d)
    L = nnz / rows
    inside = max(0.3, 1.0 - half_band_nodes / (2.0 * n_nodes))       # share of (node, offset) pairs inside the matrix
    K = max(1, int(np.ceil(1.2 * L / d / inside)))
    n_clusters = max(1, -(-K // run))
    starts = np.sort(rng.choice(np.arange(-half_band_nodes, half_band_nodes - run + 1), size=n_clusters, replace=False)) \
        if 2 * half_band_nodes - run + 1 >= n_clusters else np.arange(n_clusters) * run - (n_clusters * run) // 2
    off = np.unique(np.concatenate([starts + j for j in range(run)] + [np.zeros(1, dtype=np.int64)]))
    K = off.size
    p = min(1.0, 1.03 * L / (d * K * inside))
    for _ in range(6):
        keep = rng.random((n_nodes, K)) < p
        nz_i, nz_k = np.nonzero(keep)
        nb = nz_i + off[nz_k]
        ok = (nb >= 0) & (nb < n_nodes)
        nz_i, nb = nz_i[ok], nb[ok]
        if nz_i.size * d * d >= nnz * 1.005 or p >= 1.0:
            break
        p = min(1.0, p * 1.05 * nnz / max(1, nz_i.size * d * d))
    blocks = np.bincount(nz_i, minlength=n_nodes).astype(np.int64)          # couplings per node
    seg_start = np.concatenate([[0], np.cumsum(blocks * d)])                # per node, in the expanded column list
    E = (nb[:, None] * d + np.arange(d)[None, :]).reshape(-1)               # node's column list, ascending
    row_len = np.repeat(blocks * d, d)[:rows]
    rp = np.concatenate([[0], np.cumsum(row_len)])
    total = int(rp[-1])
    out_row = np.repeat(np.arange(rows, dtype=np.int64), row_len)
    col = E[seg_start[out_row // d] + (np.arange(total, dtype=np.int64) - rp[out_row])]
    alive = col < rows
    surplus = int(alive.sum()) - nnz
    if surplus > 0:
        u = rng.random(total)
        u[~alive] = 2.0
        thr = np.partition(u, surplus)[surplus]
        alive &= ~(u < thr)
    col = col[alive].astype(np.int32)
    out_row = out_row[alive]
    row_ptr = np.concatenate([[0], np.cumsum(np.bincount(out_row, minlength=rows))]).astype(np.int64)
    n = int(row_ptr[-1])
    val = rng.random(n, dtype=np.float32) * np.float32(2.0) - np.float32(1.0)
    val[val == 0] = np.float32(0.5)
    return row_ptr.astype(np.int32), col, val


def synth_banded(rows, cols, nnz, bandwidth, seed=0):
    return synth_csr(rows, cols, nnz, "banded", bandwidth, seed)


def make_standin(name: str, rows: int, nnz: int, fam: str, par, seed: int, uniform: bool = False):
    """One stand-in -> (row_ptr, col_idx, values, family actually used)."""
    if fam == "fem" and not uniform:
        rp, ci, va = synth_fem(rows, nnz, par[0], par[1], par[2], seed)
        return rp, ci, va, "fem"
    if fam == "fem":
        fam, par = "banded", UNIFORM_BAND[name]
    rp, ci, va = synth_csr(rows, rows, nnz, fam, par if fam == "banded" else 0, seed)
    return rp, ci, va, fam


def suitesparse_standin(name: str, uniform: bool = False):
    """-> (rows, cols, row_ptr, col_idx, values, source) for one matrix of SUITESPARSE_SET."""
    for n, rows, nnz, fam, par in SUITESPARSE_SET:
        if n == name:
            rp, ci, va, used = make_standin(name, rows, nnz, fam, par, zlib.crc32(name.encode()), uniform)
            return rows, rows, rp, ci, va, f"synthetic:{used}"
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


def _columns_for_rows(rng, lens: np.ndarray, cols: int) -> np.ndarray:
    """Ascending columns for rows of the given lengths: one uniform draw in each of len equal strata of
    [0, cols) (rows longer than cols repeat columns: duplicates are legal, spmv-helper.cpp keeps them)."""
    nnz = int(lens.sum())
    row_ptr = np.zeros(lens.size + 1, dtype=np.int64)
    np.cumsum(lens, out=row_ptr[1:])
    row_of = np.repeat(np.arange(lens.size, dtype=np.int32), lens)
    k = np.arange(nnz, dtype=np.int64) - row_ptr[:-1][row_of]
    stratum = (cols / np.maximum(lens, 1))[row_of]
    col = np.floor((k + rng.random(nnz)) * stratum).astype(np.int64)
    return np.minimum(col, cols - 1).astype(np.int32)


def _finish_csr(rng, lens: np.ndarray, cols: int):
    lens = lens.astype(np.int64)
    rp = np.zeros(lens.size + 1, dtype=np.int64)
    np.cumsum(lens, out=rp[1:])
    assert rp[-1] < 2**31
    col = _columns_for_rows(rng, lens, cols)
    val = rng.random(int(rp[-1]), dtype=np.float32) * np.float32(2.0) - np.float32(1.0)
    val[val == 0] = np.float32(0.5)
    return rp.astype(np.int32), col, val


def zipf_csr(rows: int, cols: int, nnz: int, s: float = 1.2, seed: int = 7):
    """Zipf(s) row lengths (rank k of a random row order gets ~ k^-s of the entries, capped
    at `cols`) x uniform columns.  At soc-Pokec's shape (1 632 803^2, 30.6 M) the top rows are full."""
    rng = np.random.default_rng(seed)
    w = 1.0 / (rng.permutation(rows) + 1.0) ** s
    scale = nnz / w.sum()
    for _ in range(40):                       # the cap takes entries away from the top ranks: rescale the rest
        lam = np.minimum(w * scale, cols)
        scale *= nnz / lam.sum()
    lens = np.floor(np.minimum(w * scale, cols)).astype(np.int64)
    short = int(nnz - lens.sum())
    if short > 0:
        np.add.at(lens, rng.integers(0, rows, size=short), 1)
    elif short < 0:
        cand = np.nonzero(lens > 0)[0]
        lens[rng.choice(cand, size=-short, replace=False)] -= 1
    return _finish_csr(rng, lens, cols)


def heavy_rows_csr(rows: int, cols: int, nnz: int, heavy_share: float = 0.9, heavy_rows: float = 0.01, seed: int = 11):
    """SURVEY.md 8(d) C3 adversarial: `heavy_rows` of the rows (chosen at random) hold `heavy_share` of the entries."""
    rng = np.random.default_rng(seed)
    nh = max(1, int(rows * heavy_rows))
    heavy = rng.choice(rows, size=nh, replace=False)
    lens = np.zeros(rows, dtype=np.int64)
    n_heavy = int(nnz * heavy_share)
    lens[heavy] = n_heavy // nh
    lens[heavy[: n_heavy - (n_heavy // nh) * nh]] += 1
    np.add.at(lens, rng.integers(0, rows, size=nnz - n_heavy), 1)
    return _finish_csr(rng, np.minimum(lens, 4 * cols), cols)


def full_row_plus_diagonal(n: int, row: int = 17, seed: int = 13):
    """SURVEY.md 8(d) C3 adversarial: the diagonal plus one full row (nnz of that row = cols)."""
    rng = np.random.default_rng(seed)
    lens = np.ones(n, dtype=np.int64)
    lens[row] = n
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=rp[1:])
    col = np.empty(int(rp[-1]), dtype=np.int32)
    diag = np.arange(n, dtype=np.int32)
    col[rp[:-1]] = diag
    col[rp[row]:rp[row + 1]] = diag
    val = rng.random(col.size, dtype=np.float32) * np.float32(2.0) - np.float32(1.0)
    val[val == 0] = np.float32(0.5)
    return rp.astype(np.int32), col, val


def model_test_layers(seed: int = 0):
    """SURVEY.md 8(d) C4 (apps/model_test.py:22-29, model.py:58-77 defaults): W1 dense 8192x4096, W2 sparse 8192x8192
    density 0.1, W3 sparse 1024x8192 density 0.25, with biases.  -> list of (kind, W or (r, c, v), rows, cols, bias)."""
    rng = np.random.default_rng(seed)
    out = []
    W1 = (rng.random((8192, 4096), dtype=np.float32) - np.float32(0.5)) * np.float32(0.05)
    out.append(("dense", W1, 8192, 4096, rng.random(8192, dtype=np.float32)))
    for rows, cols, dens in ((8192, 8192, 0.1), (1024, 8192, 0.25)):
        mask = rng.random((rows, cols), dtype=np.float32) < dens
        r, c = np.nonzero(mask)
        v = (rng.random(r.size, dtype=np.float32) - np.float32(0.5)) * np.float32(0.05)
        v[v == 0] = np.float32(0.01)
        out.append(("sparse", (r.astype(np.int32), c.astype(np.int32), v), rows, cols, rng.random(rows, dtype=np.float32)))
    return out


def coo_to_csr_sorted(r, c, v, rows: int):
    """COO (duplicates kept) -> CSR with rows ascending and columns ascending inside a row (stable)."""
    order = np.lexsort((c, r))
    rp = np.zeros(rows + 1, np.int64)
    np.add.at(rp, np.asarray(r, np.int64) + 1, 1)
    return np.cumsum(rp).astype(np.int32), np.asarray(c, np.int32)[order], np.asarray(v, np.float32)[order]


STANDIN_VARIANTS = ("base", "half_band", "double_band", "stray2", "stray5", "stray10", "shuffle4k")


def _resort_rows(rp, ci, va):
    """Columns ascending inside every row again (after columns were rewritten); duplicates are kept."""
    rows = rp.size - 1
    row_of = np.repeat(np.arange(rows, dtype=np.int64), np.diff(rp).astype(np.int64))
    order = np.lexsort((ci, row_of))
    return rp, ci[order], va[order]


def standin_variant(name: str, variant: str):
    """One of the eight mesh-origin (fem) stand-ins under a perturbation -- where does the launch planner fall off a cliff
    between the structured family and the pessimistic one?  (tools/standin_sweep.py)
      base                      the structured stand-in itself
      half_band / double_band   the stencil's half bandwidth x 0.5 / x 2
      stray2 / stray5 / stray10 that share of the entries re-drawn with uniform random columns (long-range couplings)
      shuffle4k                 symmetric permutation of 4096-row blocks (P A P^T: a mesh numbered block by block in random order)
    -> (rows, cols, row_ptr, col_idx, values)."""
    entry = next(q for q in SUITESPARSE_SET if q[0] == name)
    _, rows, nnz, fam, par = entry
    if fam != "fem":
        raise ValueError(f"{name} is not one of the mesh-origin stand-ins")
    seed = zlib.crc32(name.encode())
    d, run, half = par
    if variant == "half_band":
        half = max(run + 1, half // 2)
    elif variant == "double_band":
        half = min(half * 2, max(run + 1, (-(-rows // d)) - 1))
    rp, ci, va = synth_fem(rows, nnz, d, run, half, seed)
    if variant.startswith("stray"):
        share = int(variant[5:]) / 100.0
        rng = np.random.default_rng(seed + 17)
        pick = rng.random(ci.size) < share
        ci = ci.copy()
        ci[pick] = rng.integers(0, rows, size=int(pick.sum()), dtype=np.int64).astype(np.int32)
        rp, ci, va = _resort_rows(rp, ci, va)
    elif variant == "shuffle4k":
        rng = np.random.default_rng(seed + 29)
        blk = 4096
        nb = -(-rows // blk)
        perm_blocks = rng.permutation(nb)
        # new index of old row i: blocks keep their inner order, the (short) last block stays last so that sizes line up
        sizes = np.minimum(blk, rows - np.arange(nb) * blk)
        order = perm_blocks[np.argsort(perm_blocks == nb - 1, kind="stable")] if sizes[-1] != blk else perm_blocks
        new_start = np.zeros(nb, dtype=np.int64)
        new_start[order] = np.concatenate([[0], np.cumsum(sizes[order])[:-1]])
        old = np.arange(rows, dtype=np.int64)
        new_of_old = new_start[old // blk] + old % blk
        old_of_new = np.empty(rows, dtype=np.int64)
        old_of_new[new_of_old] = old
        lens = np.diff(rp).astype(np.int64)
        new_lens = lens[old_of_new]
        nrp = np.concatenate([[0], np.cumsum(new_lens)])
        src = np.repeat(rp[:-1].astype(np.int64)[old_of_new], new_lens) + (np.arange(int(nrp[-1]), dtype=np.int64) - np.repeat(nrp[:-1], new_lens))
        ci = new_of_old[ci[src]].astype(np.int32)
        va = va[src]
        rp, ci, va = _resort_rows(nrp.astype(np.int32), ci, va)
    elif variant not in ("base", "half_band", "double_band"):
        raise ValueError(variant)
    return rows, rows, rp, ci, va


def benchmark_set(names=None, uniform: bool = False, seed_shift: int = 0):
    """The matrices of a bench step (BASELINE.json configs[1]), exactly as bench.py and the parity tests build them:
    a real file under matrices/<name>/ when present, else the seeded stand-in.  -> list of dicts with
    name, source and either path or rows/cols/nnz/rp/ci/va."""
    out = []
    for name, rows, nnz, fam, par in SUITESPARSE_SET:
        if names and name not in names:
            continue
        real = real_matrix_path(name)
        if real is not None and seed_shift == 0:
            out.append(dict(name=name, source="file:" + str(real), path=str(real)))
            continue
        # HISPMV_BENCH_VARIANT=<variant of STANDIN_VARIANTS> (experiments): the mesh-origin matrices under that perturbation
        import os
        variant = os.environ.get("HISPMV_BENCH_VARIANT", "")
        if variant and fam == "fem" and not uniform and seed_shift == 0:
            _r, _c, rp, ci, va = standin_variant(name, variant)
            out.append(dict(name=name, source=f"synthetic:fem:{variant}", family=fam, rows=rows, cols=rows, nnz=int(rp[-1]), rp=rp, ci=ci, va=va))
            continue
        rp, ci, va, used = make_standin(name, rows, nnz, fam, par, zlib.crc32(name.encode()) + seed_shift, uniform)
        out.append(dict(name=name, source=f"synthetic:{used}", family=fam, rows=rows, cols=rows, nnz=int(rp[-1]), rp=rp, ci=ci, va=va))
    return out


def write_mtx(path, rows: int, cols: int, r, c, v, symmetry: str = "general", comment: str = "hispmv_amd") -> None:
    with open(path, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate real {symmetry}\n% {comment}\n{rows} {cols} {len(r)}\n")
        for i, j, x in zip(r, c, v):
            f.write(f"{int(i) + 1} {int(j) + 1} {float(x):.9g}\n")
