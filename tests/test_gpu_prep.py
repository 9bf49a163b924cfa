"""Preprocessing on the MI355X (hispmv_prep_device.hip) against the host preprocessor, which stays the
checker: COO -> CSR (stable radix sort on the device vs counting sort + row sort on the host) and CSR -> slice stream
must agree BYTE FOR BYTE -- row pointers, columns, values, every 64-bit stream word, every slice header, the split-row
list -- on the golden matrices, on inputs with duplicates / empty rows / heavy rows / unsorted entries, and at soc-Pokec's
shape; a handle created with HISPMV_PREP=device gives bit-identical y."""
import numpy as np
import pytest

from conftest import ALPHA, BETA, GOLDEN, GOLDEN_CASES, ref_vectors

pytestmark = pytest.mark.gpu

HW = ("tests.xclbin", 0, 24, 1, 1, 2, 5, True, False, True)


def assert_same(H, D, what):
    for f in ("rows", "cols", "nnz", "n_elems", "n_slices", "stream_bytes"):
        assert getattr(H, f) == getattr(D, f), (what, f)
    for f in ("row_ptr", "col_idx", "hdr", "fix"):
        assert np.array_equal(getattr(H, f), getattr(D, f)), (what, f)
    assert np.array_equal(H.values.view(np.uint32), D.values.view(np.uint32)), (what, "values")
    assert np.array_equal(H.words, D.words), (what, "words")
    assert H.plan == D.plan and np.array_equal(H.staged_words, D.staged_words), (what, "plan")


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_golden_matrices_device_stream_equals_host_stream(name):
    from hispmv_amd.prep import prep_from_coo, prep_from_coo_device, prep_from_mtx
    P = prep_from_mtx(GOLDEN / f"{name}.mtx", 1)
    r = np.repeat(np.arange(P.rows, dtype=np.int32), np.diff(P.row_ptr))
    rng = np.random.default_rng(5)
    perm = rng.permutation(r.size)                        # hand the entries over unsorted
    H = prep_from_coo(r[perm], P.col_idx[perm], P.values[perm], P.rows, P.cols)
    D, _ = prep_from_coo_device(r[perm], P.col_idx[perm], P.values[perm], P.rows, P.cols)
    assert_same(H, D, name)


@pytest.mark.parametrize("case", ["duplicates_and_empty_rows", "heavy_row", "row_aligned_short_rows", "empty_matrix", "wide_single_row"])
def test_synthetic_inputs_device_stream_equals_host_stream(case):
    from hispmv_amd.prep import prep_from_coo, prep_from_coo_device
    rng = np.random.default_rng(hash(case) % 1000)
    if case == "duplicates_and_empty_rows":
        rows, cols, nnz = 9000, 700, 120000
        r = rng.integers(0, rows, nnz); r[r % 3 == 0] = 11          # a third of the rows stay empty, one row is long
        c = rng.integers(0, 40, nnz)                                  # many duplicated (row, col) pairs: input order must survive
    elif case == "heavy_row":
        rows, cols, nnz = 3000, 500000, 900000
        r = rng.integers(0, rows, nnz); r[:600000] = 1500
        c = rng.integers(0, cols, nnz)
    elif case == "row_aligned_short_rows":
        rows, cols = 40000, 40000
        r = np.repeat(np.arange(rows), 37); nnz = r.size
        c = (r + rng.integers(-300, 300, nnz)) % cols
    elif case == "empty_matrix":
        rows, cols, nnz = 77, 33, 0
        r = np.zeros(0, np.int64); c = np.zeros(0, np.int64)
    else:
        rows, cols, nnz = 1, 200000, 150000
        r = np.zeros(nnz, np.int64); c = rng.integers(0, cols, nnz)
    v = rng.random(nnz, dtype=np.float32) - np.float32(0.5)
    if nnz:
        v[::97] = np.float32(-0.0)                                    # bit patterns travel unchanged
    H = prep_from_coo(r, c, v, rows, cols)
    D, secs = prep_from_coo_device(r, c, v, rows, cols)
    assert_same(H, D, case)
    assert all(s >= 0 for s in secs.values())


def test_out_of_range_index_is_rejected_on_the_device_too():
    from hispmv_amd.prep import prep_from_coo_device
    with pytest.raises(ValueError):
        prep_from_coo_device([0, 5], [0, 1], [1.0, 2.0], 5, 5)


def test_soc_pokec_shape_on_the_device_and_through_a_handle(monkeypatch):
    """30.6 M entries at 1 632 803^2 (the matrix whose preprocessing takes the reference 18 s): the device stream equals
    the host stream, and the handle built from it returns the same bits as one built on the host."""
    import time
    import pyhispmv
    from hispmv_amd import matrices as M
    from hispmv_amd.prep import prep_from_coo, prep_from_coo_device
    rows, _, rp, ci, va, _src = M.suitesparse_standin("soc-Pokec")
    r = np.repeat(np.arange(rows, dtype=np.int32), np.diff(rp))
    perm = np.random.default_rng(1).permutation(r.size)
    r, c, v = r[perm], ci[perm], va[perm]
    t0 = time.time(); H = prep_from_coo(r, c, v, rows, rows); t_host = time.time() - t0
    t0 = time.time(); D, secs = prep_from_coo_device(r, c, v, rows, rows); t_dev = time.time() - t0
    assert_same(H, D, "soc-Pokec shape")
    print(f"\nsoc-Pokec shape, 30.6 M unsorted COO entries: host preprocessor {t_host:.2f} s, device path {t_dev:.2f} s {secs}")
    x, b = ref_vectors(rows, rows)
    ys = []
    for mode in ("host", "device"):
        monkeypatch.setenv("HISPMV_PREP", mode)
        h = pyhispmv.FpgaHandle(*HW)
        h.set_arena_bytes(8 << 30)
        idx = h.create_sparse_handle(r, c, v, rows, rows)
        h.load_matrices()
        h.select_matrix(idx)
        y = np.zeros(rows, np.float32)
        h.run_kernel(x, b, y, ALPHA, BETA)
        ys.append(y)
        h.close()
    assert np.array_equal(ys[0].view(np.uint32), ys[1].view(np.uint32))


def _layout_cases():
    rng = np.random.default_rng(41)
    out = {}
    # a narrow band: LDS windows, every group compact
    rows = 300000
    r = np.repeat(np.arange(rows, dtype=np.int64), 14)
    out["band"] = (r, np.clip(r + rng.integers(-900, 901, r.size), 0, rows - 1), rows, rows)
    # the same band with 3 % of the entries re-drawn anywhere: compact groups with stray slots (<= 64 strays per slice)
    c = np.clip(r + rng.integers(-900, 901, r.size), 0, rows - 1)
    far = rng.random(r.size) < 0.03
    out["band_strays"] = (r, np.where(far, rng.integers(0, rows, r.size), c), rows, rows)
    # ... with 12 %: beyond the slots -- wide groups with global-column metas
    far = rng.random(r.size) < 0.12
    out["band_many_strays"] = (r, np.where(far, rng.integers(0, rows, r.size), c), rows, rows)
    # scattered columns: no window at all (wide slices, plain columns), 256-thread plan
    r2 = rng.integers(0, 50000, 400000)
    out["scattered"] = (r2, rng.integers(0, 700000, r2.size), 50000, 700000)
    # long rows (row ends rare, slices cut inside rows) over a small x
    r3 = rng.integers(0, 300, 900000)
    out["long_rows"] = (r3, rng.integers(0, 20000, r3.size), 300, 20000)
    return out


@pytest.mark.parametrize("case", ["band", "band_strays", "band_many_strays", "scattered", "long_rows"])
def test_device_layout_kernel_equals_the_host_packer(case, monkeypatch):
    """The device layout of a planned slice stream -- compact 6-byte groups, their stray slots and the stray columns behind the headers,
    wide groups -- written by layout_slices_kernel (what hispmv_load_matrices runs with HISPMV_LAYOUT=device) equals pack_device_stream byte for
    byte; and a handle loaded with either gives the same y bits."""
    import pyhispmv
    from hispmv_amd.prep import device_layout_from_coo
    r, c, rows, cols = _layout_cases()[case]
    rng = np.random.default_rng(9)
    v = rng.random(r.size, dtype=np.float32) - np.float32(0.5)
    monkeypatch.setenv("HISPMV_STRAY_SPLIT", "0")           # (the strays stay in the one stream: slots or wide groups)
    L = device_layout_from_coo(r.astype(np.int32), c.astype(np.int32), v, rows, cols, on_device=0)
    assert L["bytes"].size > 0 and np.array_equal(L["bytes"], L["bytes_device"]), case
    assert np.array_equal(L["stray_cols"], L["stray_cols_device"]), case
    if case == "band_strays":
        assert L["stray_floats"] > 0 and L["stray_slices"] > 0 and (L["stray_cols"] != 0xffffffff).any()
    if case == "band":
        assert L["compact_slices"] == L["n_slices"]
    if case in ("scattered",):
        assert L["compact_slices"] == 0
    ys = {}
    x = rng.random(cols, dtype=np.float32) - np.float32(0.3)
    b = rng.random(rows, dtype=np.float32)
    monkeypatch.setenv("HISPMV_FORMAT", "slices")
    for layout in ("device", "host"):
        monkeypatch.setenv("HISPMV_LAYOUT", layout)
        h = pyhispmv.FpgaHandle(*HW)
        idx = h.create_sparse_handle(r.astype(np.int32), c.astype(np.int32), v, rows, cols)
        h.load_matrices()
        h.select_matrix(idx)
        y = np.full(rows, np.nan, np.float32)
        h.run_kernel(x, b, y, ALPHA, BETA)
        ys[layout] = (y, h.matrix_info(idx))
        h.close()
    assert np.array_equal(ys["device"][0].view(np.uint32), ys["host"][0].view(np.uint32))
    strip = lambda d: {k: v for k, v in d.items() if k != "prep_seconds"}
    assert strip(ys["device"][1]) == strip(ys["host"][1])
