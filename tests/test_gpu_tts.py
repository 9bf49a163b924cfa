"""The transposed tile stream (hispmv_tts.h: the device format of scattered short-row matrices) on the MI355X: results
bit-identical to the CPU model of its kernel (oracle.emu_tts on the SAME packed arrays: the kernel uses no atomics, its
summation order is fixed), within the 1e-5 gate of the fp64 accumulation, identical from run to run and between the
single, batch and multi-vector entry points."""
import numpy as np
import pytest

import oracle
from conftest import ALPHA, ALPHA_HOST, BETA, BETA_HOST, TOL
from util import bwd_err

pytestmark = pytest.mark.gpu

HW = ("tests.xclbin", 0, 24, 1, 1, 2, 5, True, False, True)


def make(case, rng):
    if case == "uniform_short_rows":          # soc-Pokec-like at 1/8 size
        rows = cols = 200000
        nnz = 3600000
        r = rng.integers(0, rows, nnz); c = rng.integers(0, cols, nnz)
    elif case == "empty_and_heavy_rows":
        rows, cols, nnz = 60000, 300000, 900000
        r = rng.integers(0, rows, nnz); r[r % 4 == 0] = 31; r[:200000] = 59999      # a quarter of the rows empty, two heavy rows
        c = rng.integers(0, cols, nnz)
    elif case == "wide_columns_few_rows":     # slices cut by the 16-bit column offset
        rows, cols, nnz = 40, 3000000, 120000
        r = rng.integers(0, rows, nnz); c = rng.integers(0, cols, nnz)
    elif case == "short_wide_layer":          # x <= 256 KiB: 6 K-element tiles, four vectors share every pass of `linear`; a row cut into pieces
        rows, cols, nnz = 300, 6000, 500000
        r = rng.integers(0, rows, nnz); r[:200000] = 7
        c = rng.integers(0, cols, nnz)
    elif case == "banded_jitter":
        rows = cols = 150000
        r = np.repeat(np.arange(rows), 6); nnz = r.size
        c = (r + rng.integers(-40000, 40000, nnz)) % cols
    else:                                      # duplicates: (row, col) pairs repeated, kept as separate entries
        rows, cols, nnz = 30000, 5000, 300000
        r = rng.integers(0, rows, nnz); c = rng.integers(0, 50, nnz) * 97
    v = rng.random(r.size, dtype=np.float32) - np.float32(0.5)
    return rows, cols, r.astype(np.int32), c.astype(np.int32), v


CASES = [(c, "standard") for c in ("uniform_short_rows", "empty_and_heavy_rows", "wide_columns_few_rows", "banded_jitter", "duplicates", "short_wide_layer")] + \
        [(c, g) for g in ("tall", "paired", "zerofill", "tallgap") for c in ("uniform_short_rows", "empty_and_heavy_rows", "duplicates")]


@pytest.mark.parametrize("case,geometry", CASES)
def test_tile_stream_matches_its_model_and_the_fp64_truth(case, geometry, monkeypatch):
    """geometry "tall": two column parts of 16 K-row tiles whose absent rows have no stream word (zero-filled staging),
    part 1 through a partial vector and the merge launch, both parts pinned to XCD subsets in one grid."""
    import pyhispmv
    import torch
    from hispmv_amd.prep import prep_from_coo
    monkeypatch.setenv("HISPMV_FORMAT", "tts")
    monkeypatch.setenv("HISPMV_TTS_GEOMETRY", geometry)
    rng = np.random.default_rng(abs(hash(case)) % 997)
    rows, cols, r, c, v = make(case, rng)
    x = rng.random(cols, dtype=np.float32) - np.float32(0.3)
    b = rng.random(rows, dtype=np.float32)
    h = pyhispmv.FpgaHandle(*HW)
    idx = h.create_sparse_handle(r, c, v, rows, cols)
    h.load_matrices()
    info = h.matrix_info(idx)
    assert info["format"] == 1 and info["group_slices"] == {"standard": 28, "tall": 23, "paired": 11, "zerofill": 28, "tallgap": 23}[geometry]
    assert info["col_tiles"] == (1 if geometry in ("standard", "zerofill") else 2)
    if geometry in ("standard", "zerofill"):
        assert info["n_split_rows"] == {"empty_and_heavy_rows": 2, "short_wide_layer": 1}.get(case, 0)      # rows cut into pieces (carry tiles + fix-up)
    P = prep_from_coo(r, c, v, rows, cols, tts=(0, geometry if geometry in ("zerofill", "tallgap") else {13: 1, 28: 0, 23: "tall", 11: "paired"}[info["group_slices"]]))     # the geometry the loader chose
    rp = P.row_ptr.astype(np.int32)
    h.select_matrix(idx)
    for alpha, beta in ((ALPHA, BETA), (ALPHA_HOST, BETA_HOST), (1.0, 0.0), (-1.5, 0.5)):
        y64, mag = oracle.spmv_f64(rp, P.col_idx, P.values, x, b, alpha, beta)
        ye = oracle.emu_tts(P.tts, x, b, alpha, beta, rows)
        for _ in range(2):                      # the same bits every run
            y = np.full(rows, np.nan, np.float32)
            h.run_kernel(x, b, y, alpha, beta)
            assert bwd_err(y, y64, mag) < TOL
            assert np.array_equal(y.view(np.uint32), ye.view(np.uint32)), (case, alpha, beta)
    # multi-vector linear: several vectors per launch (small tiles: 4 / 2 vectors share every pass over the words; else one after
    # the other inside the launch), each with the bits of a single call
    X = np.concatenate([x, (x * np.float32(0.5)).astype(np.float32), x[::-1].copy(), (x + np.float32(0.25)).astype(np.float32),
                        (x * np.float32(-1.5)).astype(np.float32), np.roll(x, 17), (x * x).astype(np.float32)])
    out = h.linear(idx, X, b)
    for k in range(7):
        yk = oracle.emu_tts(P.tts, X[k * cols:(k + 1) * cols], b, 1.0, 1.0, rows)
        assert np.array_equal(out[k * rows:(k + 1) * rows].view(np.uint32), yk.view(np.uint32))
    # batch entry point, next to a slice-stream matrix and a dense handle
    dev = torch.device("cuda", 0)
    r2 = np.repeat(np.arange(4000, dtype=np.int32), 300)
    c2 = ((r2.astype(np.int64) * 7 + np.tile(np.arange(300), 4000)) % 4000).astype(np.int32)
    v2 = rng.random(r2.size, dtype=np.float32)
    i2 = h.create_sparse_handle(r2, c2, v2, 4000, 4000)
    W = rng.standard_normal((64, 96), dtype=np.float32)
    i3 = h.create_dense_handle(W.flatten(), 64, 96)
    h.load_matrices()
    assert h.matrix_info(i2)["format"] == 0
    xs = [x, rng.random(4000, dtype=np.float32), rng.random(96, dtype=np.float32)]
    bs = [b, rng.random(4000, dtype=np.float32), rng.random(64, dtype=np.float32)]
    dx = [torch.from_numpy(a).to(dev) for a in xs]
    db = [torch.from_numpy(a).to(dev) for a in bs]
    dy = [torch.full((n,), float("nan"), dtype=torch.float32, device=dev) for n in (rows, 4000, 64)]
    batch = h.prepare_batch([idx, i2, i3], [t.data_ptr() for t in dx], [t.data_ptr() for t in db], [t.data_ptr() for t in dy])
    for beta in (BETA, 0.0):
        for t in dy:
            t.fill_(float("nan"))
        torch.cuda.synchronize()
        h.spmv_device_batch(batch, ALPHA, beta)
        h.synchronize()
        ye = oracle.emu_tts(P.tts, x, b, ALPHA, beta, rows)
        assert np.array_equal(dy[0].cpu().numpy().view(np.uint32), ye.view(np.uint32))
        y2 = np.zeros(4000, np.float32)
        h.select_matrix(i2); h.run_kernel(xs[1], bs[1], y2, ALPHA, beta)
        assert np.allclose(dy[1].cpu().numpy(), y2, rtol=1e-5, atol=1e-6)
        assert np.all(np.isfinite(dy[2].cpu().numpy()))
    h.close()


def test_format_choice_follows_the_gather_locality(monkeypatch):
    """auto: matrices of >= 1 M entries with scattered short rows take the tile stream when a gather touches <= 32 lines
    of x; a matrix whose groups fit an LDS window, smaller matrices and HISPMV_FORMAT=slices keep the slice stream."""
    import pyhispmv
    rng = np.random.default_rng(3)
    monkeypatch.delenv("HISPMV_FORMAT", raising=False)
    h = pyhispmv.FpgaHandle(*HW)
    rows = cols = 750000                      # 4.5 M entries: auto considers the tile stream from 1 M
    r = np.repeat(np.arange(rows, dtype=np.int32), 6)
    c = ((r + rng.integers(-40000, 40000, r.size)) % cols).astype(np.int32)
    v = rng.random(r.size, dtype=np.float32) - np.float32(0.5)
    i_band = h.create_sparse_handle(r, c, v, rows, cols)
    rs, cs, r_s, c_s, v_s = make("banded_jitter", rng)      # the same structure below the size threshold
    i_small = h.create_sparse_handle(r_s, c_s, v_s, rs, cs)
    r2 = np.repeat(np.arange(20000, dtype=np.int32), 200)
    c2 = ((r2.astype(np.int64) + np.tile(np.arange(200), 20000)) % 20000).astype(np.int32)
    i_win = h.create_sparse_handle(r2, c2, np.ones(r2.size, np.float32), 20000, 20000)
    i_tiny = h.create_sparse_handle(r[:1000], c[:1000], v[:1000], rows, cols)
    h.load_matrices()
    a, b_, t = h.matrix_info(i_band), h.matrix_info(i_win), h.matrix_info(i_tiny)
    assert a["format"] == 1 and 0 < a["tts_lines_per_gather"] <= 32
    assert h.matrix_info(i_small)["format"] == 0
    assert b_["format"] == 0 and b_["lds_bytes"] > 0
    assert t["format"] == 0
    h.close()
    monkeypatch.setenv("HISPMV_FORMAT", "slices")
    h = pyhispmv.FpgaHandle(*HW)
    i = h.create_sparse_handle(r, c, v, rows, cols)
    h.load_matrices()
    assert h.matrix_info(i)["format"] == 0
    h.close()
