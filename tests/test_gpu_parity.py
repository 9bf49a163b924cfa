"""Parity of the HIP path (through the C ABI / FpgaHandle facade) against the oracle, on a real
MI355X.  Tolerance: BASELINE.json north_star -- indices bit-exact (checked on the host in
test_prep_host.py), y within 1e-5 relative fp32, evaluated in backward-error form
|dy_i| <= 1e-5 * (|alpha| sum_j |a_ij x_j| + |beta b_i|) against an fp64 accumulation."""
import numpy as np
import pytest

import oracle
from conftest import ALPHA, ALPHA_HOST, BETA, BETA_HOST, GOLDEN, GOLDEN_CASES, TOL, ref_vectors
from util import bwd_err

pytestmark = pytest.mark.gpu

HW = ("tests.xclbin", 0, 24, 1, 1, 2, 5, True, False, True)   # apps/general_test.py:10-19


@pytest.fixture(autouse=True)
def slice_stream_only(monkeypatch):
    """This module checks the SLICE stream bit for bit against its wavefront model; matrices that would take the
    transposed tile stream (scattered short rows) are covered by tests/test_gpu_tts.py and test_gpu_bench_set.py."""
    monkeypatch.setenv("HISPMV_FORMAT", "slices")


@pytest.fixture(scope="module")
def pyhispmv_mod():
    import pyhispmv
    return pyhispmv


@pytest.fixture()
def fpga(pyhispmv_mod):
    h = pyhispmv_mod.FpgaHandle(*HW)
    yield h
    h.close()


def prepared_tiles(info, r, c, v, rows, cols):
    """The slice streams of the handle's column tiles, packed on the host the way the loader packs them."""
    from hispmv_amd.prep import prep_from_coo
    r, c, v = np.asarray(r), np.asarray(c), np.asarray(v, np.float32)
    if info["col_tiles"] <= 1:
        return [prep_from_coo(r, c, v, rows, cols)]
    if info.get("tile_kind") == 3:
        # stray split: part 0 = the entries inside the x window of their workgroup (under the plan of the WHOLE matrix), part 1 the rest
        from hispmv_amd.prep import window_membership
        inside, order = window_membership(r, c, v, rows, cols, 256)
        keep = np.zeros(r.size, dtype=bool)
        keep[order] = inside.astype(bool)
        return [prep_from_coo(r[keep], c[keep], v[keep], rows, cols), prep_from_coo(r[~keep], c[~keep], v[~keep], rows, cols)]
    width, base, n = info["col_tile_width"], info["col_tile_base"], info["col_tiles"]
    # tile_kind 2 (band tiles): base / width are ranges of the OFFSET from the scaled diagonal, col - row*cols/rows
    key = c.astype(np.int64) - (r.astype(np.int64) * cols // rows) if info.get("tile_kind") == 2 else c.astype(np.int64)
    tiles = []
    for t in range(n):                                     # the end tiles are open-ended (hispmv.h: col_tile_base)
        lo = -(1 << 40) if t == 0 else base + t * width
        hi = (1 << 40) if t == n - 1 else base + (t + 1) * width
        sel = (key >= lo) & (key < hi)
        tiles.append(prep_from_coo(r[sel], c[sel], v[sel], rows, cols))
    return tiles


def emulate_tiles(tiles, x, b, alpha, beta, rows, mode):
    """Tile 0 computes alpha*A_0*x + beta*bias; every further column tile computes alpha*A_t*x into a partial vector
    (its own cut rows fixed up there), and the merge pass adds the partial vectors to y in tile order."""
    ye = None
    for t, P in enumerate(tiles):
        if t == 0:
            ye = oracle.emu_spmv(P.words, P.hdr, P.fix, x, b, alpha, beta, rows, mode)
        else:
            ye = (ye + oracle.emu_spmv(P.words, P.hdr, P.fix, x, np.zeros(rows, np.float32), alpha, 0.0, rows, mode)).astype(np.float32)
    return ye


def emulate_device(info, r, c, v, rows, cols, x, b, alpha, beta, carry=None):
    """The wavefront model applied the way the device runs the handle: one stream per column tile (tile 0
    with beta*bias, later tiles into partial vectors added afterwards), carry variant as reported by matrix_info (or `carry`:
    a batched pass always uses the fix-up variant, 0)."""
    return emulate_tiles(prepared_tiles(info, r, c, v, rows, cols), x, b, alpha, beta, rows,
                         info["carry_lookback"] if carry is None else carry)


def csr_truth(r, c, v, rows, x, b, alpha, beta):
    order = np.lexsort((c, r))
    rp = np.zeros(rows + 1, np.int64)
    np.add.at(rp, np.asarray(r, np.int64) + 1, 1)
    rp = np.cumsum(rp).astype(np.int32)
    return oracle.spmv_f64(rp, np.asarray(c, np.int32)[order], np.asarray(v, np.float32)[order], x, b, alpha, beta)


def test_native_library_is_loaded(pyhispmv_mod):
    from hispmv_amd import _lib
    assert _lib.LIB_PATH.exists()
    maps = open("/proc/self/maps").read()
    assert "libhispmv.so" in maps


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_golden_matrices_vs_mkl_and_emulator(name, golden, fpga):
    from hispmv_amd.prep import prep_from_mtx
    g = golden(name)
    idx = fpga.create_sparse_handle_from_mtx(GOLDEN / f"{name}.mtx", 1)
    assert idx == 0
    fpga.load_matrices()
    fpga.select_matrix(idx)
    rows, cols = int(g["rows"]), int(g["cols"])
    x, y0 = ref_vectors(rows, cols)
    y = np.full(rows, np.nan, np.float32)
    fpga.run_kernel(x, y0, y, ALPHA, BETA)
    y64, mag = oracle.spmv_f64(g["ref_row_ptr"], g["ref_col_idx"], g["ref_vals"], x, y0, ALPHA, BETA)
    assert bwd_err(y, y64, mag) < TOL
    assert float(np.max(np.abs(y.astype(np.float64) - g["y_mkl"]) / mag)) < 3e-6        # vs the cpu/ MKL path (measured envelope: 1.4e-6)
    pl, mre, _ = oracle.precision_loss(g["y_mkl"], y)                                    # the reference's own metric
    assert pl < 1e-5
    # the CPU model of the wavefront performs the same fp32 operations in the same order
    P = prep_from_mtx(GOLDEN / f"{name}.mtx", 1)
    rr = np.repeat(np.arange(rows, dtype=np.int32), np.diff(P.row_ptr))
    ye = emulate_device(fpga.matrix_info(idx), rr, P.col_idx, P.values, rows, cols, x, y0, ALPHA, BETA)
    assert np.array_equal(y.view(np.uint32), ye.view(np.uint32)), "GPU result differs bitwise from the wavefront model"
    info = fpga.matrix_info(idx)
    assert info["nnz"] == g["ref_col_idx"].size and info["loaded"] == 1 and not info["is_dense"]


@pytest.mark.parametrize("carry,mode", [("lookback", 1), ("fixup", 0), ("resident", None), ("auto", None)])
def test_both_carry_variants_match_their_wavefront_model(pyhispmv_mod, monkeypatch, carry, mode):
    """Rows shared between slices: the two-launch fix-up variant (default) and the single-launch
    look-back (HISPMV_CARRY=lookback) each reproduce their CPU model bit for bit, on a matrix with
    short chains, a 100-slice chain and empty rows."""
    from hispmv_amd.prep import prep_from_coo
    monkeypatch.setenv("HISPMV_CARRY", carry)
    rng = np.random.default_rng(3)
    rows, cols, nnz = 20000, 15000, 400000
    r = rng.integers(0, rows, nnz)
    r[:100000] = 1234
    r[r % 5 == 0] += 1
    c = rng.integers(0, cols, nnz)
    v = rng.random(nnz, dtype=np.float32) - 0.5
    x = rng.random(cols, dtype=np.float32)
    b = rng.random(rows, dtype=np.float32)
    h = pyhispmv_mod.FpgaHandle(*HW)
    idx = h.create_sparse_handle(r, c, v, rows, cols)
    h.load_matrices()
    h.select_matrix(idx)
    P = prep_from_coo(r, c, v, rows, cols)
    info = h.matrix_info(idx)
    if mode is not None:
        assert info["carry_lookback"] == mode
    y64, mag = oracle.spmv_f64(P.row_ptr.astype(np.int32), P.col_idx, P.values, x, b, ALPHA, BETA)
    ye = emulate_device(info, r, c, v, rows, cols, x, b, ALPHA, BETA)
    for _ in range(3):                      # repeated launches reuse the granules with a new launch tag
        y = np.full(rows, np.nan, np.float32)
        h.run_kernel(x, b, y, ALPHA, BETA)
        assert bwd_err(y, y64, mag) < TOL
        assert np.array_equal(y.view(np.uint32), ye.view(np.uint32))
    h.close()


def test_column_tiled_scattered_matrix(fpga):
    """x larger than 1.5x the column-tile budget (4 MiB) with scattered columns: the matrix is cut into column
    tiles (the reference's column tiling, spmv-helper.cpp:242-263); tile 0 applies beta*bias, later tiles
    accumulate in place.  Bit-exact against the wavefront model applied tile by tile."""
    from hispmv_amd.prep import prep_from_coo
    rng = np.random.default_rng(13)
    rows, cols, nnz = 150000, 1800000, 1500000
    r = rng.integers(0, rows, nnz).astype(np.int32)
    c = rng.integers(0, cols, nnz).astype(np.int32)
    v = rng.random(nnz, dtype=np.float32) - 0.5
    x = rng.random(cols, dtype=np.float32)
    b = rng.random(rows, dtype=np.float32)
    idx = fpga.create_sparse_handle(r, c, v, rows, cols)
    fpga.load_matrices()
    info = fpga.matrix_info(idx)
    assert info["col_tiles"] == 2 and info["lds_bytes"] == 0 and info["col_tile_width"] % 64 == 0
    y = np.zeros(rows, np.float32)
    fpga.select_matrix(idx)
    fpga.run_kernel(x, b, y, ALPHA, BETA)
    y64, mag = csr_truth(r, c, v, rows, x, b, ALPHA, BETA)
    assert bwd_err(y, y64, mag) < TOL
    assert np.array_equal(y.view(np.uint32), emulate_device(info, r, c, v, rows, cols, x, b, ALPHA, BETA).view(np.uint32))
    # the same block inside a four times wider x (a rank's shard of a larger matrix, x replicated at full length):
    # tiled over the columns it uses, not over the width of x -- still two tiles, none of them empty
    off, wide = 2 * cols + 1000, 4 * cols
    xw = rng.random(wide, dtype=np.float32)
    c2 = c + off
    c2[:40] = rng.integers(0, wide, 40)                     # a few strays anywhere (a shard's rows of the next block)
    idx2 = fpga.create_sparse_handle(r, c2, v, rows, wide)
    fpga.load_matrices()
    info2 = fpga.matrix_info(idx2)
    assert info2["col_tiles"] == 2 and abs(info2["col_tile_base"] - off) < 8192
    fpga.select_matrix(idx2)
    fpga.run_kernel(xw, b, y, ALPHA, BETA)
    y64, mag = csr_truth(r, c2, v, rows, xw, b, ALPHA, BETA)
    assert bwd_err(y, y64, mag) < TOL
    assert np.array_equal(y.view(np.uint32), emulate_device(info2, r, c2, v, rows, wide, xw, b, ALPHA, BETA).view(np.uint32))


@pytest.mark.parametrize("split", [0, 1])
def test_window_of_most_used_blocks_with_l2_spill(pyhispmv_mod, monkeypatch, split):
    """split = 0 (HISPMV_STRAY_SPLIT=0, the plan of rounds 1-3): x too large for one LDS window (88 % of the entries in 30 000 hot columns, the others anywhere in 400 000; column
    tiling switched off so that the single-stream plan runs): the group's window holds its most used 64-byte blocks and
    the remaining elements gather through L2 in the same pass.  Also a banded matrix with a few far-away couplings per row (blocks used
    once or twice stay out of the window).  Bit-exact against the wavefront model.
    split = 1 (default since round 4): 12 % / 8 % of the entries outside the windows -> the stray split (tile_kind 3): the windowed
    part with clean windows, the strays through L2 into a partial vector; bit-exact against the model of the two parts."""
    monkeypatch.setenv("HISPMV_COL_TILE_BYTES", "0")
    monkeypatch.setenv("HISPMV_STRAY_SPLIT", str(split))
    fpga = pyhispmv_mod.FpgaHandle(*HW)
    rng = np.random.default_rng(21)
    cases = []
    rows, cols = 16000, 400000
    r = np.repeat(np.arange(rows, dtype=np.int32), 500)
    cases.append((rows, cols, r, np.where(rng.random(r.size) < 0.88, rng.integers(0, 30000, r.size), rng.integers(30000, cols, r.size)).astype(np.int32)))
    rows, cols = 200000, 200000
    r = np.repeat(np.arange(rows, dtype=np.int32), 24)
    c = (r + rng.integers(-300, 300, r.size)) % cols
    far = rng.random(r.size) < 0.08                                   # 8 % of the couplings go anywhere
    c = np.where(far, rng.integers(0, cols, r.size), c).astype(np.int32)
    cases.append((rows, cols, r, c))
    for rows, cols, r, c in cases:
        v = rng.random(r.size, dtype=np.float32) - 0.5
        x = rng.random(cols, dtype=np.float32)
        b = rng.random(rows, dtype=np.float32)
        idx = fpga.create_sparse_handle(r, c, v, rows, cols)
        fpga.load_matrices()
        info = fpga.matrix_info(idx)
        if split:
            assert info["tile_kind"] == 3 and info["col_tiles"] == 2 and info["lds_bytes"] > 0, info
        else:
            assert info["col_tiles"] == 1 and info["lds_bytes"] > 0, info
        y = np.zeros(rows, np.float32)
        fpga.select_matrix(idx)
        fpga.run_kernel(x, b, y, ALPHA, BETA)
        y64, mag = csr_truth(r, c, v, rows, x, b, ALPHA, BETA)
        assert bwd_err(y, y64, mag) < TOL
        assert np.array_equal(y.view(np.uint32), emulate_device(info, r, c, v, rows, cols, x, b, ALPHA, BETA, carry=0 if split else None).view(np.uint32))
    fpga.close()


def test_general_test_call_sequence_scaled(fpga):
    """apps/general_test.py:22-113 with the same call order, dense 5000x1000 + 100 k random COO
    (duplicates included), checked with the script's own np.allclose(rtol=1e-3) and the 1e-5 gate."""
    from scipy.sparse import coo_matrix
    np.random.seed(0)
    rows, cols = 5000, 1000
    dense_values = np.random.rand(rows, cols).astype(np.float32)
    x = np.random.rand(cols).astype(np.float32)
    bias = np.random.rand(rows).astype(np.float32)
    y_dense = np.zeros(rows, dtype=np.float32)
    y_dense_expected = np.dot(dense_values, x) + bias
    nnz = 100000
    coo_rows = np.random.randint(0, rows, size=nnz, dtype=np.int32)
    coo_cols = np.random.randint(0, cols, size=nnz, dtype=np.int32)
    coo_values = np.random.rand(nnz).astype(np.float32)
    y_sparse = np.zeros(rows, dtype=np.float32)
    y_sparse_expected = coo_matrix((coo_values, (coo_rows, coo_cols)), shape=(rows, cols)).dot(x) + bias
    dense_handle_idx = fpga.create_dense_handle(dense_values.flatten(), rows, cols)
    sparse_handle_idx = fpga.create_sparse_handle(coo_rows, coo_cols, coo_values, rows, cols)
    assert (dense_handle_idx, sparse_handle_idx) == (0, 1)
    fpga.load_matrices()
    fpga.select_matrix(dense_handle_idx)
    fpga.run_kernel(x, bias, y_dense, 1.0, 1.0)
    fpga.select_matrix(sparse_handle_idx)
    fpga.run_kernel(x, bias, y_sparse, 1.0, 1.0)
    assert np.allclose(y_dense, y_dense_expected, rtol=1e-3)
    assert np.allclose(y_sparse, y_sparse_expected, rtol=1e-3)
    y64, mag = csr_truth(coo_rows, coo_cols, coo_values, rows, x, bias, 1.0, 1.0)
    assert bwd_err(y_sparse, y64, mag) < TOL
    d64 = dense_values.astype(np.float64) @ x.astype(np.float64) + bias
    assert bwd_err(y_dense, d64, np.abs(dense_values.astype(np.float64)) @ np.abs(x.astype(np.float64)) + np.abs(bias)) < TOL
    assert np.array_equal(y_dense.view(np.uint32), oracle.emu_gemv(dense_values, x, bias, 1.0, 1.0).view(np.uint32))


def test_model_test_call_sequence_scaled(fpga):
    """apps/model_test.py + fpga_layer_manager.py:54-81 + model.py:68-80: dense layer, two sparse
    layers (densities 0.1 / 0.25), ReLU between, `linear(idx, x_flat, bias)` per layer."""
    rng = np.random.default_rng(0)
    sizes = [(1024, 512, None), (1024, 1024, 0.1), (256, 1024, 0.25)]     # (out, in, density)
    Ws, bs, idxs = [], [], []
    for out_f, in_f, dens in sizes:
        W = rng.standard_normal((out_f, in_f), dtype=np.float32)
        if dens is not None:
            W *= rng.random((out_f, in_f)) < dens
        b = rng.standard_normal(out_f, dtype=np.float32)
        density = np.count_nonzero(W) / W.size
        if density > 0.5:                                                  # fpga_layer_manager.py:43-47
            idx = fpga.create_dense_handle(W.flatten(), *W.shape)
        else:
            rr, cc = np.nonzero(W)
            idx = fpga.create_sparse_handle(rr.astype(np.int64), cc.astype(np.int64), W[rr, cc], *W.shape)   # int64 like torch
        assert idx >= 0
        Ws.append(W); bs.append(b); idxs.append(idx)
    fpga.load_matrices()
    x = rng.standard_normal(512, dtype=np.float32)
    h, h64 = x, x.astype(np.float64)
    for W, b, idx in zip(Ws, bs, idxs):
        y = fpga.linear(idx, h, b)
        assert y.shape == (W.shape[0],)
        y64 = W.astype(np.float64) @ h.astype(np.float64) + b
        mag = np.abs(W.astype(np.float64)) @ np.abs(h.astype(np.float64)) + np.abs(b)
        assert bwd_err(y, y64, mag) < TOL
        h = np.maximum(y, 0)
    # batched linear: num_vecs = len(x)//cols, same bias for every vector (fpga_handle.cpp:336-339)
    xb = rng.standard_normal(3 * 512, dtype=np.float32)
    yb = fpga.linear(idxs[0], xb, bs[0])
    assert yb.shape == (3 * 1024,)
    for k in range(3):
        assert np.array_equal(yb[k * 1024:(k + 1) * 1024], fpga.linear(idxs[0], xb[k * 512:(k + 1) * 512], bs[0]))


def test_batched_linear_shares_one_pass_over_the_matrix(fpga):
    """linear() with several vectors (fpga_handle.cpp:336-379 runs the kernel once per vector): here 8/4/2 (dense)
    or 4/2 (sparse) vectors share one pass.  Dense: bitwise equal to the single-vector call.  Sparse: every vector
    bitwise equal to the wavefront model with the fix-up carry variant -- LDS-window plan, L2-gather plan,
    short rows with look-back as single-vector default, and a column-tiled matrix (per-vector bias = y)."""
    rng = np.random.default_rng(33)
    W = rng.standard_normal((700, 1001), dtype=np.float32)               # odd column count: dword path
    W4 = rng.standard_normal((300, 512), dtype=np.float32)
    b_d, b_d4 = rng.standard_normal(700, dtype=np.float32), rng.standard_normal(300, dtype=np.float32)
    i_d, i_d4 = fpga.create_dense_handle(W.flatten(), *W.shape), fpga.create_dense_handle(W4.flatten(), *W4.shape)
    sparse = []
    for rows, cols, nnz, kind in [(4096, 4096, 1700000, "lds window"), (30000, 20000, 300000, "l2 gathers"),
                                  (200000, 200000, 900000, "short rows"), (150000, 1800000, 1500000, "column tiles")]:
        r = rng.integers(0, rows, nnz).astype(np.int32)
        c = rng.integers(0, cols, nnz).astype(np.int32)
        if kind == "short rows":
            c = ((r.astype(np.int64) + rng.integers(-50, 50, nnz)) % cols).astype(np.int32)
        v = rng.random(nnz, dtype=np.float32) - 0.5
        sparse.append((fpga.create_sparse_handle(r, c, v, rows, cols), rows, cols, r, c, v, kind))
    fpga.load_matrices()
    for idx, Wd, bd in ((i_d, W, b_d), (i_d4, W4, b_d4)):
        rows, cols = Wd.shape
        for nv in (2, 3, 8, 13):
            xb = rng.standard_normal(nv * cols, dtype=np.float32)
            yb = fpga.linear(idx, xb, bd)
            assert yb.shape == (nv * rows,)
            for k in range(nv):
                assert np.array_equal(yb[k * rows:(k + 1) * rows], fpga.linear(idx, xb[k * cols:(k + 1) * cols], bd))
    for idx, rows, cols, r, c, v, kind in sparse:
        info = fpga.matrix_info(idx)
        if kind == "lds window":
            assert info["lds_bytes"] > 0
        if kind == "column tiles":
            assert info["col_tiles"] == 2
        b = rng.standard_normal(rows, dtype=np.float32)
        tiles = prepared_tiles(info, r, c, v, rows, cols)
        for nv in (2, 4, 7):
            xb = rng.standard_normal(nv * cols, dtype=np.float32)
            yb = fpga.linear(idx, xb, b)
            for k in range(nv):
                xk = xb[k * cols:(k + 1) * cols]
                y64, mag = csr_truth(r, c, v, rows, xk, b, 1.0, 1.0)
                assert bwd_err(yb[k * rows:(k + 1) * rows], y64, mag) < TOL, (kind, nv, k)
                # `linear` takes the fix-up carry variant for every vector, batched pass or the odd last one (one vector or
                # many, a vector of a `linear` call always has the same bits)
                ye = emulate_tiles(tiles, xk, b, 1.0, 1.0, rows, 0)
                assert np.array_equal(yb[k * rows:(k + 1) * rows].view(np.uint32), ye.view(np.uint32)), (kind, nv, k)


def test_multi_matrix_launch_matches_each_matrix_alone(fpga):
    """hispmv_spmv_device_batch: several independent matrices share launches (one grid per workgroup size, one fix-up
    launch, column tiles in rounds).  Every y must equal, bit for bit, the wavefront model of that matrix with the
    fix-up carry variant -- LDS-window plans, L2-gather plans, short rows (look-back when run alone), a column-tiled
    matrix, an empty matrix and a dense handle in the same call."""
    import torch
    rng = np.random.default_rng(44)
    specs = [(4096, 4096, 1700000, "uniform"), (30000, 20000, 300000, "uniform"), (200000, 200000, 900000, "band"),
             (150000, 1800000, 1500000, "uniform"), (64, 64, 0, "uniform"), (9000, 9000, 700000, "band"), (500, 40000, 260000, "uniform")]
    mats = []
    for rows, cols, nnz, kind in specs:
        r = rng.integers(0, rows, nnz).astype(np.int32)
        c = rng.integers(0, cols, nnz).astype(np.int32)
        if kind == "band":
            c = ((r.astype(np.int64) * cols // rows + rng.integers(-40, 40, nnz)) % cols).astype(np.int32)
        v = rng.random(nnz, dtype=np.float32) - 0.5
        mats.append(dict(idx=fpga.create_sparse_handle(r, c, v, rows, cols), rows=rows, cols=cols, r=r, c=c, v=v))
    Wd = rng.standard_normal((300, 520), dtype=np.float32)
    i_dense = fpga.create_dense_handle(Wd.flatten(), *Wd.shape)
    fpga.load_matrices()
    dev = torch.device("cuda", 0)
    for m in mats:
        m["x"] = rng.random(m["cols"], dtype=np.float32); m["b"] = rng.random(m["rows"], dtype=np.float32)
        m["dx"], m["db"] = torch.from_numpy(m["x"]).to(dev), torch.from_numpy(m["b"]).to(dev)
        m["dy"] = torch.full((m["rows"],), float("nan"), dtype=torch.float32, device=dev)
    xd, bd = rng.random(520, dtype=np.float32), rng.random(300, dtype=np.float32)
    dxd, dbd, dyd = torch.from_numpy(xd).to(dev), torch.from_numpy(bd).to(dev), torch.zeros(300, device=dev)
    batch = fpga.prepare_batch([m["idx"] for m in mats] + [i_dense], [m["dx"].data_ptr() for m in mats] + [dxd.data_ptr()],
                               [m["db"].data_ptr() for m in mats] + [dbd.data_ptr()], [m["dy"].data_ptr() for m in mats] + [dyd.data_ptr()])
    for m in mats:
        m["tiles"] = prepared_tiles(fpga.matrix_info(m["idx"]), m["r"], m["c"], m["v"], m["rows"], m["cols"])
    for alpha, beta in ((ALPHA, BETA), (1.0, 0.0)):
        for m in mats:
            m["ye"] = emulate_tiles(m["tiles"], m["x"], m["b"], alpha, beta, m["rows"], 0)
            m["y64"], m["mag"] = csr_truth(m["r"], m["c"], m["v"], m["rows"], m["x"], m["b"], alpha, beta)
        for _ in range(2):                   # the second call reuses the cached device tables
            for m in mats:
                m["dy"].fill_(float("nan"))
            fpga.spmv_device_batch(batch, alpha, beta)
            fpga.synchronize()
            torch.cuda.synchronize()
            for m in mats:
                y = m["dy"].cpu().numpy()
                assert bwd_err(y, m["y64"], m["mag"]) < TOL
                assert np.array_equal(y.view(np.uint32), m["ye"].view(np.uint32)), (m["rows"], m["cols"], alpha, beta)
            yd = np.zeros(300, np.float32)
            fpga.select_matrix(i_dense)
            fpga.run_kernel(xd, bd, yd, alpha, beta)
            assert np.array_equal(dyd.cpu().numpy(), yd)
    m0, m1 = mats[0], mats[1]
    with pytest.raises(ValueError):          # one SpMV per sparse handle at a time: its carry buffers are shared
        fpga.spmv_device_batch(fpga.prepare_batch([m0["idx"], m0["idx"]], [m0["dx"].data_ptr()] * 2, [m0["db"].data_ptr()] * 2,
                                                  [m0["dy"].data_ptr(), m1["dy"].data_ptr()]), 1.0, 1.0)
    with pytest.raises(ValueError):          # two results in the same place
        fpga.spmv_device_batch(fpga.prepare_batch([m0["idx"], m1["idx"]], [m0["dx"].data_ptr(), m1["dx"].data_ptr()],
                                                  [m0["db"].data_ptr(), m1["db"].data_ptr()], [m0["dy"].data_ptr()] * 2), 1.0, 1.0)


def test_dense_handles_of_a_batch_share_one_grid(fpga):
    """hispmv_spmv_device_batch with several dense overlay handles (the sizes of cpu/run_gemv.sh:9-13 scaled down, a row
    count that is not a multiple of the 4-row block, a width that is not a multiple of 4, the same handle twice): one
    GeMV grid for all of them, largest first -- every y bit-identical to the handle launched alone and to the CPU model,
    with and without bias, next to a sparse matrix in the same call."""
    import torch
    rng = np.random.default_rng(77)
    dev = torch.device("cuda", 0)
    shapes = [(64, 64), (1024, 1024), (301, 520), (2048, 512), (17, 4099), (1024, 1024)]
    ds = []
    for rows, cols in shapes:
        W = rng.standard_normal((rows, cols), dtype=np.float32)
        ds.append(dict(W=W, rows=rows, cols=cols, idx=fpga.create_dense_handle(W.flatten(), rows, cols)))
    ds[5]["idx"] = ds[1]["idx"]; ds[5]["W"] = ds[1]["W"]            # the same dense handle twice (different vectors)
    rs = rng.integers(0, 20000, 400000).astype(np.int32); cs = rng.integers(0, 20000, 400000).astype(np.int32)
    vs = rng.random(400000, dtype=np.float32) - 0.5
    i_sp = fpga.create_sparse_handle(rs, cs, vs, 20000, 20000)
    fpga.load_matrices()
    for d in ds:
        d["x"], d["b"] = rng.random(d["cols"], dtype=np.float32), rng.random(d["rows"], dtype=np.float32)
        d["dx"], d["db"] = torch.from_numpy(d["x"]).to(dev), torch.from_numpy(d["b"]).to(dev)
        d["dy"] = torch.full((d["rows"],), float("nan"), dtype=torch.float32, device=dev)
    xs, bs = rng.random(20000, dtype=np.float32), rng.random(20000, dtype=np.float32)
    dxs, dbs, dys = torch.from_numpy(xs).to(dev), torch.from_numpy(bs).to(dev), torch.zeros(20000, device=dev)
    batch = fpga.prepare_batch([d["idx"] for d in ds[:3]] + [i_sp] + [d["idx"] for d in ds[3:]],
                               [d["dx"].data_ptr() for d in ds[:3]] + [dxs.data_ptr()] + [d["dx"].data_ptr() for d in ds[3:]],
                               [d["db"].data_ptr() for d in ds[:3]] + [dbs.data_ptr()] + [d["db"].data_ptr() for d in ds[3:]],
                               [d["dy"].data_ptr() for d in ds[:3]] + [dys.data_ptr()] + [d["dy"].data_ptr() for d in ds[3:]])
    for alpha, beta in ((ALPHA, BETA), (1.0, 0.0)):
        for _ in range(2):
            for d in ds:
                d["dy"].fill_(float("nan"))
            fpga.spmv_device_batch(batch, alpha, beta)
            fpga.synchronize()
            torch.cuda.synchronize()
            for d in ds:
                y = d["dy"].cpu().numpy()
                alone = np.zeros(d["rows"], np.float32)
                fpga.select_matrix(d["idx"])
                fpga.run_kernel(d["x"], d["b"], alone, alpha, beta)
                assert np.array_equal(y.view(np.uint32), alone.view(np.uint32)), (d["rows"], d["cols"], alpha, beta)
                assert np.array_equal(y.view(np.uint32), oracle.emu_gemv(d["W"], d["x"], d["b"], alpha, beta).view(np.uint32))
            y64, mag = csr_truth(rs, cs, vs, 20000, xs, bs, alpha, beta)
            assert bwd_err(dys.cpu().numpy(), y64, mag) < TOL


def test_csr_with_unsorted_rows_and_null_arrays(fpga):
    """hispmv_create_sparse_handle_from_csr: rows whose columns are not ascending (scipy: has_sorted_indices == False) are
    sorted on the way in -- same bits as the sorted matrix --, NULL column / value arrays with entries are rejected
    (everything downstream takes a row's first and last column from its ends)."""
    import ctypes as C
    from hispmv_amd._lib import lib
    rng = np.random.default_rng(21)
    rows, cols, nnz = 3000, 2500, 90000
    r = np.sort(rng.integers(0, rows, nnz)).astype(np.int32)
    c = rng.integers(0, cols, nnz).astype(np.int32)                 # unsorted inside the rows
    v = rng.random(nnz, dtype=np.float32) - np.float32(0.5)
    rp = np.zeros(rows + 1, np.int64)
    np.add.at(rp, r.astype(np.int64) + 1, 1)
    rp = np.cumsum(rp).astype(np.int32)
    order = np.lexsort((c, r))                                      # stable inside equal (row, col)
    i_uns = fpga.create_sparse_handle_from_csr(rp, c, v, rows, cols)
    i_srt = fpga.create_sparse_handle_from_csr(rp, c[order], v[order], rows, cols)
    fpga.load_matrices()
    x, b = rng.random(cols, dtype=np.float32), rng.random(rows, dtype=np.float32)
    ys = []
    for i in (i_uns, i_srt):
        y = np.zeros(rows, np.float32)
        fpga.select_matrix(i)
        fpga.run_kernel(x, b, y, ALPHA, BETA)
        ys.append(y)
    assert np.array_equal(ys[0].view(np.uint32), ys[1].view(np.uint32))
    y64, mag = oracle.spmv_f64(rp, c[order], v[order], x, b, ALPHA, BETA)
    assert bwd_err(ys[0], y64, mag) < TOL
    rc = lib.hispmv_create_sparse_handle_from_csr(fpga._ctx, C.c_void_p(rp.ctypes.data), None, None, rows, cols)
    assert rc == -2                                                  # HISPMV_EINVAL


def test_batch_tables_survive_the_first_host_vector_call(pyhispmv_mod):
    """Regression (round 1, hispmv_abi.cpp run_host_vectors): the first staged run_kernel / linear of a context grew the
    pinned staging block and, through a stray line, freed the device tables of earlier hispmv_spmv_device_batch calls
    without forgetting them -- the next batch call with the same arguments launched on freed memory.  Sequence: batch,
    first host-vector call (grows the stage), a few small device allocations that would reuse the freed fragments and
    overwrite them, the same batch again: bit-identical results, and no free rejected by the runtime up to and
    including hispmv_destroy."""
    import torch
    from hispmv_amd._lib import lib
    rng = np.random.default_rng(9)
    dev = torch.device("cuda", 0)
    before = lib.hispmv_free_failures()
    h = pyhispmv_mod.FpgaHandle(*HW)
    mats = []
    for rows, cols, nnz in ((5000, 4000, 90000), (12000, 12000, 300000), (300, 70000, 40000)):
        r = rng.integers(0, rows, nnz).astype(np.int32)
        r[: nnz // 5] = 7                                   # cut rows: the fix-up table is used too
        c = rng.integers(0, cols, nnz).astype(np.int32)
        v = rng.random(nnz, dtype=np.float32) - 0.5
        mats.append(dict(idx=h.create_sparse_handle(r, c, v, rows, cols), rows=rows, cols=cols))
    Wd = rng.standard_normal((64, 128), dtype=np.float32)
    i_dense = h.create_dense_handle(Wd.flatten(), *Wd.shape)
    h.load_matrices()
    for m in mats:
        m["dx"] = torch.from_numpy(rng.random(m["cols"], dtype=np.float32)).to(dev)
        m["db"] = torch.from_numpy(rng.random(m["rows"], dtype=np.float32)).to(dev)
        m["dy"] = torch.full((m["rows"],), float("nan"), dtype=torch.float32, device=dev)
    batch = h.prepare_batch([m["idx"] for m in mats], [m["dx"].data_ptr() for m in mats], [m["db"].data_ptr() for m in mats],
                            [m["dy"].data_ptr() for m in mats])
    h.spmv_device_batch(batch, ALPHA, BETA)
    h.synchronize()
    first = [m["dy"].cpu().numpy().copy() for m in mats]
    yd = np.zeros(64, np.float32)
    h.select_matrix(i_dense)
    h.run_kernel(rng.random(128, dtype=np.float32), rng.random(64, dtype=np.float32), yd, 1.0, 1.0)   # grows the stage
    import ctypes as C
    hip = C.CDLL(None)                                       # the HIP runtime already in the process (torch's copy)
    junk = []
    if hasattr(hip, "hipMalloc"):
        for _ in range(64):                                  # raw small allocations: they take freed fragments first
            p = C.c_void_p()
            assert hip.hipMalloc(C.byref(p), C.c_size_t(2048)) == 0
            assert hip.hipMemset(p, C.c_int(0xff), C.c_size_t(2048)) == 0
            junk.append(p)
        assert hip.hipDeviceSynchronize() == 0
    for m in mats:
        m["dy"].fill_(float("nan"))
    torch.cuda.synchronize()
    h.spmv_device_batch(batch, ALPHA, BETA)                  # same key: the cached tables
    h.synchronize()
    for m, y0 in zip(mats, first):
        assert np.array_equal(m["dy"].cpu().numpy().view(np.uint32), y0.view(np.uint32))
    for p in junk:
        assert hip.hipFree(p) == 0
    h.close()
    assert lib.hispmv_free_failures() == before


@pytest.mark.parametrize("alpha,beta", [(ALPHA_HOST, BETA_HOST), (1.0, 0.0), (0.0, 1.0), (-1.5, 0.5)])
def test_power_law_and_heavy_rows(fpga, alpha, beta):
    rng = np.random.default_rng(7)
    rows, cols, nnz = 60000, 50000, 1500000
    w = 1.0 / (np.arange(rows) + 1.0) ** 1.2
    r = rng.choice(rows, size=nnz, p=w / w.sum()).astype(np.int32)          # Zipf row lengths, many empty rows
    c = rng.integers(0, cols, nnz).astype(np.int32)
    v = (rng.random(nnz, dtype=np.float32) - 0.5)
    idx = fpga.create_sparse_handle(r, c, v, rows, cols)
    fpga.load_matrices()
    info = fpga.matrix_info(idx)
    assert info["n_split_rows"] > 0 and info["n_elems"] > info["nnz"]        # shared rows and empty-row fillers exist
    x = rng.random(cols, dtype=np.float32)
    b = rng.random(rows, dtype=np.float32) if beta != 0 else np.full(rows, np.nan, np.float32)
    y = np.zeros(rows, np.float32)
    fpga.select_matrix(idx)
    fpga.run_kernel(x, b, y, alpha, beta)
    y64, mag = csr_truth(r, c, v, rows, x, np.nan_to_num(b), alpha, beta)
    assert np.isfinite(y).all() and bwd_err(y, y64, mag) < TOL
    y2 = np.zeros(rows, np.float32)
    fpga.run_kernel(x, b, y2, alpha, beta)
    assert np.array_equal(y.view(np.uint32), y2.view(np.uint32))             # no atomics: bitwise reproducible


def test_adversarial_full_row_plus_diagonal(fpga):
    n = 300000
    r = np.concatenate([np.arange(n), np.full(n, 12345)]).astype(np.int32)
    c = np.concatenate([np.arange(n), np.arange(n)]).astype(np.int32)
    rng = np.random.default_rng(1)
    v = rng.random(2 * n, dtype=np.float32) + 0.5
    idx = fpga.create_sparse_handle(r, c, v, n, n)
    fpga.load_matrices()
    assert fpga.matrix_info(idx)["n_split_rows"] >= 1
    x, y0 = ref_vectors(n, n)
    y = np.zeros(n, np.float32)
    fpga.select_matrix(idx)
    fpga.run_kernel(x, y0, y, ALPHA, BETA)
    y64, mag = csr_truth(r, c, v, n, x, y0, ALPHA, BETA)
    assert bwd_err(y, y64, mag) < TOL


def test_full_size_suitesparse_shape_properties(fpga):
    """A BASELINE-size stand-in (crankseg_2 shape: 63 838 rows, 14.1 M nnz) checked through
    size-independent properties: linearity in x, A*1 = row sums, alpha/beta scaling."""
    from hispmv_amd.matrices import synth_banded
    rows, nnz = 63838, 14148850
    rp, ci, va = synth_banded(rows, rows, nnz, bandwidth=4000, seed=2)
    idx = fpga.create_sparse_handle_from_csr(rp, ci, va, rows, rows)
    fpga.load_matrices()
    fpga.select_matrix(idx)
    ones = np.ones(rows, np.float32)
    zero = np.zeros(rows, np.float32)
    y1 = np.zeros(rows, np.float32)
    fpga.run_kernel(ones, zero, y1, 1.0, 0.0)
    rowsum = np.add.reduceat(va.astype(np.float64), rp[:-1].astype(np.int64)) * (np.diff(rp) > 0)
    mag = np.add.reduceat(np.abs(va.astype(np.float64)), rp[:-1].astype(np.int64)) * (np.diff(rp) > 0)
    assert bwd_err(y1, rowsum, np.maximum(mag, 1e-30)) < TOL
    rng = np.random.default_rng(4)
    xa = rng.integers(-8, 9, rows).astype(np.float32)
    xb = rng.integers(-8, 9, rows).astype(np.float32)
    ya, yb, yab = (np.zeros(rows, np.float32) for _ in range(3))
    fpga.run_kernel(xa, zero, ya, 1.0, 0.0)
    fpga.run_kernel(xb, zero, yb, 1.0, 0.0)
    fpga.run_kernel(xa + xb, zero, yab, 1.0, 0.0)
    scale = np.maximum(mag * 16, 1e-30)
    assert float(np.max(np.abs(yab.astype(np.float64) - ya - yb) / scale)) < 3 * TOL
    y2 = np.zeros(rows, np.float32)
    fpga.run_kernel(xa, ones, y2, 2.0, 3.0)
    assert float(np.max(np.abs(y2.astype(np.float64) - (2.0 * ya.astype(np.float64) + 3.0)) / np.maximum(scale, 3.0))) < 3 * TOL
    y64, m64 = oracle.spmv_f64(rp, ci, va, xa, zero, 1.0, 0.0)
    assert bwd_err(ya, y64, np.maximum(m64, 1e-30)) < TOL


def test_dense_shapes_and_unaligned_columns(fpga):
    rng = np.random.default_rng(9)
    shapes = [(1, 1), (3, 7), (130, 1001), (257, 4096), (1000, 2500)]
    idxs = []
    mats = []
    for rws, cls in shapes:
        W = rng.random((rws, cls), dtype=np.float32) - 0.5
        idxs.append(fpga.create_dense_handle(W.flatten(), rws, cls))
        mats.append(W)
    fpga.load_matrices()
    for idx, W in zip(idxs, mats):
        rws, cls = W.shape
        x = rng.random(cls, dtype=np.float32)
        b = rng.random(rws, dtype=np.float32)
        y = np.zeros(rws, np.float32)
        fpga.select_matrix(idx)
        fpga.run_kernel(x, b, y, ALPHA, BETA)
        ref = ALPHA * (W.astype(np.float64) @ x.astype(np.float64)) + BETA * b.astype(np.float64)
        mag = abs(ALPHA) * (np.abs(W.astype(np.float64)) @ np.abs(x.astype(np.float64))) + np.abs(BETA * b.astype(np.float64))
        assert bwd_err(y, ref, mag) < TOL
        assert np.array_equal(y.view(np.uint32), oracle.emu_gemv(W, x, b, ALPHA, BETA).view(np.uint32))
        yn = oracle.naive_gemv(W, x, b, ALPHA, BETA)           # cpu/src/main.cpp:53-71
        assert bwd_err(yn, ref, mag) < TOL


def test_capacity_and_error_contract(pyhispmv_mod):
    h = pyhispmv_mod.FpgaHandle(*HW)
    h.set_arena_bytes(1 << 20)
    W = np.ones((600, 600), np.float32)                          # 1.44 MB > 1 MiB
    assert h.create_dense_handle(W.flatten(), 600, 600) == -1    # fpga_handle.cpp:235-238
    small = h.create_dense_handle(W[:100, :100].flatten(), 100, 100)
    assert small == 0
    r = np.arange(200000, dtype=np.int32) % 1000
    assert h.create_sparse_handle(r, r, np.ones(r.size, np.float32), 1000, 1000) == -1   # :192-195
    with pytest.raises(AssertionError):
        h.run_kernel(np.ones(100, np.float32), np.ones(100, np.float32), np.zeros(100, np.float32), 1.0, 1.0)  # nothing selected (:292)
    with pytest.raises(IndexError):
        h.select_matrix(5)                                       # :267-270
    h.select_matrix(small)
    with pytest.raises(AssertionError):                          # not loaded yet
        h.run_kernel(np.ones(100, np.float32), np.ones(100, np.float32), np.zeros(100, np.float32), 1.0, 1.0)
    h.load_matrices()
    h.load_matrices()                                            # idempotent (reference: corrupts offsets, :259-261)
    y = np.zeros(100, np.float32)
    h.run_kernel(np.ones(100, np.float32), np.ones(100, np.float32), y, 1.0, 1.0)
    assert np.all(y == 101.0)
    with pytest.raises(TypeError):
        h.run_kernel(np.ones(100, np.float32), np.ones(100, np.float32), np.zeros(100, np.float64), 1.0, 1.0)
    with pytest.raises(ValueError):
        h.create_sparse_handle([0, 2000], [0, 0], [1.0, 1.0], 1000, 1000)   # index outside the matrix
    h.close()
    nd = pyhispmv_mod.FpgaHandle("t.xclbin", 0, 16, 1, 4, 2, 5, False, False, True)   # no dense overlay
    with pytest.raises(AssertionError):
        nd.create_dense_handle(np.ones(4, np.float32), 2, 2)        # spmv-helper.cpp:718
    nd.close()
    with pytest.raises(RuntimeError):
        pyhispmv_mod.FpgaHandle("t.xclbin", 64, 24, 1, 1, 2, 5, True, False, True)   # no such device


def test_device_resident_entry_point_matches_host_path(fpga):
    import torch
    rng = np.random.default_rng(11)
    rows, cols, nnz = 20000, 20000, 300000
    r = rng.integers(0, rows, nnz).astype(np.int32)
    c = rng.integers(0, cols, nnz).astype(np.int32)
    v = rng.random(nnz, dtype=np.float32) - 0.5
    idx = fpga.create_sparse_handle(r, c, v, rows, cols)
    fpga.load_matrices()
    x = rng.random(cols, dtype=np.float32)
    b = rng.random(rows, dtype=np.float32)
    y = np.zeros(rows, np.float32)
    fpga.select_matrix(idx)
    fpga.run_kernel(x, b, y, ALPHA, BETA)
    dx, db = torch.from_numpy(x).cuda(), torch.from_numpy(b).cuda()
    dy = torch.zeros(rows, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    fpga.spmv_device(idx, dx.data_ptr(), db.data_ptr(), dy.data_ptr(), ALPHA, BETA, stream)
    torch.cuda.synchronize()
    assert np.array_equal(dy.cpu().numpy().view(np.uint32), y.view(np.uint32))
    ms = fpga.time_device(idx, dx.data_ptr(), db.data_ptr(), dy.data_ptr(), ALPHA, BETA, 20)
    assert 0 < ms < 50


_LINEAR_BITS = {}


def oracle_linear(case, v, W, x, b):
    """fp64 y = A x + b and the magnitude sum of its terms, for a COO case or the dense W."""
    r, c, nr, nc = case
    if r is None:
        return W.astype(np.float64) @ x.astype(np.float64) + b, np.abs(W).astype(np.float64) @ np.abs(x).astype(np.float64) + np.abs(b)
    t = np.zeros(nr); m = np.zeros(nr)
    p = v.astype(np.float64) * x.astype(np.float64)[c]
    np.add.at(t, r, p); np.add.at(m, r, np.abs(p))
    return t + b, m + np.abs(b)


@pytest.mark.parametrize("mode", ["direct", "copy"])
def test_host_vector_path_direct_and_copied_y(pyhispmv_mod, mode, monkeypatch):
    """run_kernel / linear from host buffers: by default x and bias reach the device through a fetch kernel that reads the pinned
    staging block and y is written straight into that block (no DMA copies: hispmv_abi.cpp run_host_vectors); HISPMV_HOST_Y=copy keeps
    the copies.  Both give the bits of the device entry point -- for a slice stream with cut rows (the tail's read-modify-writes of y
    cross PCIe), a matrix in two column parts (partial vector + merge), a dense handle, 5 vectors per linear call, beta = 0, and a
    matrix whose vectors are too large for the staging block (> 8 MB: plain copies in both modes)."""
    import torch
    monkeypatch.setenv("HISPMV_HOST_Y", mode)
    h = pyhispmv_mod.FpgaHandle(*HW)
    rng = np.random.default_rng(5)
    cases = []
    rows, cols, nnz = 30000, 26000, 900000                      # long rows: many rows cut by slice boundaries
    r = rng.integers(0, rows // 50, nnz).astype(np.int32) * 50; c = rng.integers(0, cols, nnz).astype(np.int32)
    cases.append((r, c, rows, cols))
    rows2 = 6000; cols2 = 60000                                # x of 240 KB on short rows: two LDS-window column tiles
    r2 = np.repeat(np.arange(rows2, dtype=np.int32), 700); c2 = rng.integers(0, cols2, r2.size).astype(np.int32)
    cases.append((r2, c2, rows2, cols2))
    rows3 = cols3 = 1200000                                     # 4.8 MB vectors: x + bias + y exceed the staging threshold
    r3 = rng.integers(0, rows3, 2000000).astype(np.int32); c3 = rng.integers(0, cols3, r3.size).astype(np.int32)
    cases.append((r3, c3, rows3, cols3))
    idxs, vals = [], []
    for (rr, cc, nr, nc) in cases:
        vals.append(rng.random(rr.size, dtype=np.float32) - np.float32(0.5))
        idxs.append(h.create_sparse_handle(rr, cc, vals[-1], nr, nc))
    vals.append(None)
    W = rng.standard_normal((300, 700), dtype=np.float32)
    idxs.append(h.create_dense_handle(W.flatten(), 300, 700))
    cases.append((None, None, 300, 700))
    h.load_matrices()
    dev = torch.device("cuda", 0)
    for idx, (_, _, nr, nc) in zip(idxs, cases):
        x = rng.random(nc, dtype=np.float32) - np.float32(0.3)
        b = rng.random(nr, dtype=np.float32)
        dx, db = torch.from_numpy(x).to(dev), torch.from_numpy(b).to(dev)
        h.select_matrix(idx)
        for alpha, beta in ((ALPHA, BETA), (1.25, 0.0)):
            dy = torch.full((nr,), float("nan"), dtype=torch.float32, device=dev)
            h.spmv_device(idx, dx.data_ptr(), db.data_ptr(), dy.data_ptr(), alpha, beta, None)
            h.synchronize()
            for _ in range(2):
                y = np.full(nr, np.nan, np.float32)
                h.run_kernel(x, b, y, alpha, beta)
                assert np.array_equal(y.view(np.uint32), dy.cpu().numpy().view(np.uint32)), (mode, idx, alpha, beta)
        if nr * 5 * 4 < (4 << 20):
            X = rng.random(5 * nc, dtype=np.float32)
            out = h.linear(idx, X, b)
            y64 = [oracle_linear(cases[idxs.index(idx)], vals[idxs.index(idx)], W, X[k * nc:(k + 1) * nc], b) for k in range(5)]
            for k in range(5):
                t, mag = y64[k]
                assert bwd_err(out[k * nr:(k + 1) * nr], t, mag) < TOL
            _LINEAR_BITS.setdefault(idx, out.copy())            # (the same seeds in both modes: the second one must reproduce the first)
            assert np.array_equal(out.view(np.uint32), _LINEAR_BITS[idx].view(np.uint32)), (mode, idx)
    h.close()


def test_batch_layout_of_short_groups(pyhispmv_mod, monkeypatch):
    """A resident plan whose groups come out short (here 6.9 M entries: 27 slices per workgroup, nd6k's case) gets a second device
    layout with groups two to four times as long (hispmv_matrix_info.batch_group_slices) that hispmv_spmv_device_batch takes when the call
    shares the chip (two lanes: >= 256 MiB of streams -- forced here with HISPMV_BATCH_STREAMS=2); single launches keep the first
    plan.  Same slices, same carries: the batch call's y equals the single launch's y bit for bit, with the layout on and off, next
    to a large matrix and a tile stream in the same call."""
    import torch
    monkeypatch.setenv("HISPMV_BATCH_STREAMS", "2")
    rng = np.random.default_rng(21)
    rows = cols = 18000
    per = 383                                                     # nd6k's density: 6.9 M entries in a band of +-2500
    r = np.repeat(np.arange(rows, dtype=np.int64), per)
    c = np.clip(r + rng.integers(-2500, 2501, r.size), 0, cols - 1).astype(np.int32)
    r = r.astype(np.int32)
    v = rng.random(r.size, dtype=np.float32) - np.float32(0.5)
    rows2 = cols2 = 100000                                        # a second, larger slice matrix (groups of > 80 slices: no batch layout)
    r2 = np.repeat(np.arange(rows2, dtype=np.int64), 300)
    c2 = np.clip(r2 + rng.integers(-1500, 1501, r2.size), 0, cols2 - 1).astype(np.int32)
    r2 = r2.astype(np.int32)
    v2 = rng.random(r2.size, dtype=np.float32) - np.float32(0.5)
    dev = torch.device("cuda", 0)
    x, b = rng.random(cols, dtype=np.float32) - np.float32(0.3), rng.random(rows, dtype=np.float32)
    x2, b2 = rng.random(cols2, dtype=np.float32) - np.float32(0.3), rng.random(rows2, dtype=np.float32)
    results = {}
    for layout in ("1", "0"):
        monkeypatch.setenv("HISPMV_BATCH_LAYOUT", layout)
        h = pyhispmv_mod.FpgaHandle(*HW)
        i1 = h.create_sparse_handle(r, c, v, rows, cols)
        i2 = h.create_sparse_handle(r2, c2, v2, rows2, cols2)
        h.load_matrices()
        info1, info2 = h.matrix_info(i1), h.matrix_info(i2)
        assert info1["block_threads"] == 1024 and info1["group_slices"] < 40 and info1["lds_bytes"] > 0, info1
        if layout == "1":       # four times as long where that plan keeps its kind, else three times, else twice
            assert any(abs(info1["batch_group_slices"] - k * info1["group_slices"]) <= k for k in (2, 3, 4)), info1
        else:
            assert info1["batch_group_slices"] == 0, info1
        assert info2["batch_group_slices"] == 0 and info2["group_slices"] >= 80, info2
        dx, db, dx2, db2 = (torch.from_numpy(a).to(dev) for a in (x, b, x2, b2))
        dy = torch.full((rows,), float("nan"), dtype=torch.float32, device=dev)
        dy2 = torch.full((rows2,), float("nan"), dtype=torch.float32, device=dev)
        h.spmv_device(i1, dx.data_ptr(), db.data_ptr(), dy.data_ptr(), ALPHA, BETA, None)
        h.synchronize()
        single = dy.cpu().numpy().copy()
        batch = h.prepare_batch([i2, i1], [dx2.data_ptr(), dx.data_ptr()], [db2.data_ptr(), db.data_ptr()], [dy2.data_ptr(), dy.data_ptr()])
        for _ in range(2):
            dy.fill_(float("nan")); dy2.fill_(float("nan"))
            torch.cuda.synchronize()
            h.spmv_device_batch(batch, ALPHA, BETA, None)
            h.synchronize()
            assert np.array_equal(dy.cpu().numpy().view(np.uint32), single.view(np.uint32)), layout
        P = prepared_tiles(info1, r, c, v, rows, cols)[0]
        y64, mag = oracle.spmv_f64(P.row_ptr.astype(np.int32), P.col_idx, P.values, x, b, ALPHA, BETA)
        assert bwd_err(single, y64, mag) < TOL
        results[layout] = (single, dy2.cpu().numpy().copy())
        h.close()
    assert np.array_equal(results["1"][0].view(np.uint32), results["0"][0].view(np.uint32))
    assert np.array_equal(results["1"][1].view(np.uint32), results["0"][1].view(np.uint32))


def test_wide_band_is_cut_along_the_diagonal(fpga):
    """Band tiles (hispmv_matrix_info.tile_kind 2): a banded matrix whose band (+-24000 here) is wider than an LDS window is cut
    into ranges of the offset from the diagonal; every part then runs with its x window in LDS and 6-byte elements.  Bitwise
    equal to the wavefront model of the parts (part 0 with the bias, the others through partial vectors), single launch and
    batch entry point; within the 1e-5 gate of the fp64 truth."""
    import torch
    rng = np.random.default_rng(77)
    rows = cols = 400000
    per_row, half = 12, 24000
    r = np.repeat(np.arange(rows, dtype=np.int64), per_row)
    c = np.clip(r + rng.integers(-half, half + 1, r.size), 0, cols - 1)
    v = rng.random(r.size, dtype=np.float32) - np.float32(0.5)
    r, c = r.astype(np.int32), c.astype(np.int32)
    x = rng.random(cols, dtype=np.float32) - np.float32(0.3)
    b = rng.random(rows, dtype=np.float32)
    idx = fpga.create_sparse_handle(r, c, v, rows, cols)
    fpga.load_matrices()
    info = fpga.matrix_info(idx)
    assert info["format"] == 0 and info["tile_kind"] == 2 and info["col_tiles"] >= 2 and info["lds_bytes"] > 0
    assert info["compact_slices"] >= 0.9 * info["n_slices"]
    y64, mag = csr_truth(r, c, v, rows, x, b, ALPHA, BETA)
    ye = emulate_device(info, r, c, v, rows, cols, x, b, ALPHA, BETA, carry=0)
    fpga.select_matrix(idx)
    for _ in range(2):
        y = np.full(rows, np.nan, np.float32)
        fpga.run_kernel(x, b, y, ALPHA, BETA)
        assert bwd_err(y, y64, mag) < TOL
        assert np.array_equal(y.view(np.uint32), ye.view(np.uint32))
    dev = torch.device("cuda", 0)
    dx, db = torch.from_numpy(x).to(dev), torch.from_numpy(b).to(dev)
    dy = torch.full((rows,), float("nan"), dtype=torch.float32, device=dev)
    batch = fpga.prepare_batch([idx], [dx.data_ptr()], [db.data_ptr()], [dy.data_ptr()])
    fpga.spmv_device_batch(batch, ALPHA, BETA)
    fpga.synchronize()
    assert np.array_equal(dy.cpu().numpy().view(np.uint32), ye.view(np.uint32))


@pytest.mark.parametrize("slots", [1, 0, 2])
def test_stray_couplings(pyhispmv_mod, monkeypatch, slots):
    """A banded matrix with 3 % of its entries at random columns: every workgroup has a few elements outside its x window -- unsplit
    and without stray slots all of them would take 8-byte elements and the two-way gather (the cliff tools/standin_sweep.py found:
    PFlow_742 0.70 -> 0.43 of the roofline at 2 % strays).
    slots = 1 (default): STRAY SLOTS (hispmv_plan.h) -- every slice stays compact, the owning wavefront fetches its slice's strays
    (<= 64) into its stray area behind the window one slice ahead; one stream, tile_kind 0.
    slots = 0 (HISPMV_STRAY_SLOTS=0): the STRAY SPLIT (tile_kind 3) -- windowed part + strays through L2 into a partial vector.
    Both: bitwise equal to the wavefront model, single launch, batch entry point and `linear` with 3 vectors; within the 1e-5 gate."""
    import torch
    monkeypatch.setenv("HISPMV_STRAY_SLOTS", str(min(slots, 1)))
    if slots == 2:
        # the look-back variant walks a group in slice order, the packer placed the strays for the rotated walk of the fix-up variant:
        # a matrix with stray slots keeps the fix-up variant whatever HISPMV_CARRY asks for
        monkeypatch.setenv("HISPMV_CARRY", "lookback")
    fpga = pyhispmv_mod.FpgaHandle(*HW)
    try:
        rng = np.random.default_rng(91)
        rows = cols = 300000
        per_row, half = 16, 1500
        r = np.repeat(np.arange(rows, dtype=np.int64), per_row)
        c = np.clip(r + rng.integers(-half, half + 1, r.size), 0, cols - 1)
        stray = rng.random(r.size) < 0.03
        c[stray] = rng.integers(0, cols, int(stray.sum()))
        v = rng.random(r.size, dtype=np.float32) - np.float32(0.5)
        r, c = r.astype(np.int32), c.astype(np.int32)
        x = rng.random(cols, dtype=np.float32) - np.float32(0.3)
        b = rng.random(rows, dtype=np.float32)
        idx = fpga.create_sparse_handle(r, c, v, rows, cols)
        fpga.load_matrices()
        info = fpga.matrix_info(idx)
        tiles = prepared_tiles(info, r, c, v, rows, cols)
        if slots:
            assert info["format"] == 0 and info["tile_kind"] == 0 and info["col_tiles"] == 1 and info["lds_bytes"] > 0, info
            assert info["compact_slices"] == info["n_slices"]                   # 6-byte elements everywhere, strays included
        else:
            assert info["format"] == 0 and info["tile_kind"] == 3 and info["col_tiles"] == 2 and info["lds_bytes"] > 0, info
            assert tiles[0].nnz + tiles[1].nnz == r.size and 0.02 * r.size < tiles[1].nnz < 0.05 * r.size
        y64, mag = csr_truth(r, c, v, rows, x, b, ALPHA, BETA)
        assert not (slots and info["carry_lookback"])
        ye = emulate_tiles(tiles, x, b, ALPHA, BETA, rows, 0)
        fpga.select_matrix(idx)
        for _ in range(2):
            y = np.full(rows, np.nan, np.float32)
            fpga.run_kernel(x, b, y, ALPHA, BETA)
            assert bwd_err(y, y64, mag) < TOL
            assert np.array_equal(y.view(np.uint32), ye.view(np.uint32))
        dev = torch.device("cuda", 0)
        dx, db = torch.from_numpy(x).to(dev), torch.from_numpy(b).to(dev)
        dy = torch.full((rows,), float("nan"), dtype=torch.float32, device=dev)
        batch = fpga.prepare_batch([idx], [dx.data_ptr()], [db.data_ptr()], [dy.data_ptr()])
        fpga.spmv_device_batch(batch, ALPHA, BETA)
        fpga.synchronize()
        yb = dy.cpu().numpy()
        assert bwd_err(yb, y64, mag) < TOL
        assert np.array_equal(yb.view(np.uint32), emulate_tiles(tiles, x, b, ALPHA, BETA, rows, 0).view(np.uint32))
        # linear: 3 vectors in one call (alpha = beta = 1), each with the bits of its own one-vector call
        xs = rng.random(3 * cols, dtype=np.float32)
        out = fpga.linear(idx, xs, b)
        for k in range(3):
            yk64, mk = csr_truth(r, c, v, rows, xs[k * cols:(k + 1) * cols], b, 1.0, 1.0)
            assert bwd_err(out[k * rows:(k + 1) * rows], yk64, mk) < TOL
        one = fpga.linear(idx, xs[:cols], b)
        assert np.array_equal(one.view(np.uint32), out[:rows].view(np.uint32))
    finally:
        fpga.close()


def test_strays_without_slots_and_without_split_take_the_two_way_gather(pyhispmv_mod, monkeypatch):
    monkeypatch.setenv("HISPMV_STRAY_SPLIT", "0")
    monkeypatch.setenv("HISPMV_STRAY_SLOTS", "0")
    rng = np.random.default_rng(92)
    rows = cols = 200000
    r = np.repeat(np.arange(rows, dtype=np.int64), 16)
    c = np.clip(r + rng.integers(-1500, 1501, r.size), 0, cols - 1)
    stray = rng.random(r.size) < 0.03
    c[stray] = rng.integers(0, cols, int(stray.sum()))
    v = rng.random(r.size, dtype=np.float32) - np.float32(0.5)
    h = pyhispmv_mod.FpgaHandle(*HW)
    try:
        idx = h.create_sparse_handle(r.astype(np.int32), c.astype(np.int32), v, rows, cols)
        h.load_matrices()
        info = h.matrix_info(idx)
        assert info["tile_kind"] == 0 and info["col_tiles"] == 1 and info["compact_slices"] < 0.1 * info["n_slices"]
        x = rng.random(cols, dtype=np.float32)
        b = rng.random(rows, dtype=np.float32)
        y = np.full(rows, np.nan, np.float32)
        h.select_matrix(idx)
        h.run_kernel(x, b, y, ALPHA, BETA)
        y64, mag = csr_truth(r, c, v, rows, x, b, ALPHA, BETA)
        assert bwd_err(y, y64, mag) < TOL
        assert np.array_equal(y.view(np.uint32), emulate_device(info, r.astype(np.int32), c.astype(np.int32), v, rows, cols, x, b, ALPHA, BETA).view(np.uint32))
    finally:
        h.close()
