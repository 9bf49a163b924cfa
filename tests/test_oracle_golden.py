"""The oracle against the reference's known answers and the committed golden vectors (CPU only).

Pins (DESIGN.md "Oracle"): KAT-0 / KAT-1 are the known-answer values of SURVEY.md Appendix C.2;
ref_* arrays in tests/golden/*.npz are outputs of the reference's own compiled cpu/ loader;
y_mkl is MKL called the way cpu/src/main.cpp does."""
import hashlib

import numpy as np
import pytest

import oracle
from conftest import ALPHA, BETA, GOLDEN, GOLDEN_CASES, TOL, ref_vectors
from util import bwd_err, kat0_coo


def test_kat0_reference_packer_words():
    rows, cols, r, c, v = kat0_coo()
    h = oracle.RefPack(16, 1, 4, 2, 5, False, False, True).prepare_sparse(rows, cols, r, c, v)
    info = h.info()
    assert info["words_per_channel"] == 2112 and info["run_length"] == 264
    assert info["row_tiles"] == 1 and info["col_tiles"] == 1 and info["rows_per_pe"] == 8
    assert h.count_bit(46) == 3840          # sharedRow words
    assert h.count_bit(63) == 21426         # valid words
    assert h.hash() == 0x2FD5EA5BEF2F27B9   # FNV-1a-64 over all channels' words
    y = oracle.cpu_sequential(r, c, v, rows, np.ones(cols, np.float32), np.zeros(rows, np.float32), 1.0, 0.0)
    assert abs(float(y[7]) - 1011.403198) < 5e-4
    assert f"{float(y[7]):.6f}" == "1011.403198"


def test_kat1_syn1138_file_and_cpu_driver_numbers(golden):
    mtx = GOLDEN / "syn_1138.mtx"
    assert hashlib.md5(mtx.read_bytes()).hexdigest() == "f57a3087e03217f667a21fbeae60405d"
    rows, cols, cr, cc, cv = oracle.load_mtx_common(mtx)
    assert (rows, cols, cr.size) == (1138, 1138, 4054)
    # reference quirk (spmv-helper.cpp:92): without the trailing newline the last entry is lost
    g = golden("syn_1138")
    h = oracle.RefPack(24, 1, 1, 2, 5, True, False, True).prepare_sparse(rows, cols, cr, cc, cv)
    info = h.info()
    assert (info["padded_rows"], info["padded_cols"], info["rows_per_pe"], info["run_length"]) == (1152, 1152, 6, 60)
    pl, mre, mi = oracle.precision_loss(g["y_cpu_spmv"], g["y_mkl"])
    assert f"{mre:.6f}" == "0.000007" and mi == 1101 and f"{pl:.6f}" == "0.000000"


def test_common_loader_drops_last_line_without_newline(tmp_path):
    src = (GOLDEN / "syn_1138.mtx").read_bytes()
    p = tmp_path / "nonl.mtx"
    p.write_bytes(src.rstrip(b"\n"))
    assert oracle.load_mtx_common(p)[2].size == 4053


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_cpu_loader_restatement_matches_reference_output(name, golden):
    g = golden(name)
    rows, cols, rp, ci, va = oracle.read_mtx_cpu(GOLDEN / f"{name}.mtx")
    assert (rows, cols) == (int(g["rows"]), int(g["cols"]))
    assert np.array_equal(rp, g["ref_row_ptr"]) and np.array_equal(ci, g["ref_col_idx"])
    assert np.array_equal(va.view(np.uint32), g["ref_vals"].view(np.uint32))


@pytest.mark.skipif(not oracle.ref_available(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_golden_equals_live_reference_build(name, golden):
    g = golden(name)
    rows, cols, rp, ci, va = oracle.ref_read_mtx_csr(GOLDEN / f"{name}.mtx")
    assert np.array_equal(rp, g["ref_row_ptr"]) and np.array_equal(ci, g["ref_col_idx"]) and np.array_equal(va, g["ref_vals"])


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_cpu_spmv_restatement_vs_mkl_and_fixture(name, golden):
    g = golden(name)
    rows, cols = int(g["rows"]), int(g["cols"])
    x, y0 = ref_vectors(rows, cols)
    y = oracle.cpu_spmv(g["ref_row_ptr"], g["ref_col_idx"], g["ref_vals"], x, y0, ALPHA, BETA, 1)
    assert np.array_equal(y, g["y_cpu_spmv"])
    y64, mag = oracle.spmv_f64(g["ref_row_ptr"], g["ref_col_idx"], g["ref_vals"], x, y0, ALPHA, BETA)
    assert bwd_err(y, y64, mag) < TOL
    assert bwd_err(g["y_mkl"], y64, mag) < TOL


def test_loader_flavours_differ_where_the_reference_does(golden):
    g = golden("skew")
    assert g["coo_r"].size == 2 * g["ref_col_idx"].size      # common/ mirrors skew (negated), cpu/ does not
    g = golden("gen_real")
    assert g["ref_col_idx"].size == g["coo_r"].size + 1      # cpu/ keeps the "-0.0" entry, common/ drops it


HW = [  # (num_ch_A, num_ch_B, num_ch_C, urams, fp_acc_latency, dense, pre_acc, row_dist)
    (24, 1, 1, 2, 5, True, False, True),    # apps tuple
    (16, 1, 4, 2, 5, False, False, True),
    (16, 2, 4, 2, 4, False, True, True),
    (24, 1, 1, 2, 4, False, True, False),
    (2, 1, 1, 1, 5, False, False, True),    # small PE count -> multi-tile
]


@pytest.mark.parametrize("hw", HW)
def test_reference_stream_emulator_reproduces_cpu_sequential(hw):
    rng = np.random.default_rng(hash(hw) % 2**32)
    rows, cols, nnz = 70000 if hw[0] == 2 else 3000, 20000, 60000
    r = rng.integers(0, rows, nnz).astype(np.int32)
    r[: nnz // 8] = 11                                   # one heavy row -> shared rows
    c = rng.integers(0, cols, nnz).astype(np.int32)
    v = (rng.random(nnz, dtype=np.float32) - 0.5)
    x = rng.random(cols, dtype=np.float32)
    cin = rng.random(rows, dtype=np.float32)
    h = oracle.RefPack(*hw).prepare_sparse(rows, cols, r, c, v)
    info = h.info()
    if hw[0] == 2:
        assert info["row_tiles"] > 1 and info["col_tiles"] > 1
    y = h.emulate(x, cin, 0.55, -2.05)
    yc = oracle.cpu_sequential(r, c, v, rows, x, cin, 0.55, -2.05)
    # both are fp32 sums in different orders; compare each to fp64 in backward-error form
    order = np.lexsort((c, r))
    rp = np.zeros(rows + 1, np.int32)
    np.add.at(rp, r + 1, 1)
    rp = np.cumsum(rp).astype(np.int32)
    y64, mag = oracle.spmv_f64(rp, c[order], v[order], x, cin, 0.55, -2.05)
    assert bwd_err(y, y64, mag) < TOL and bwd_err(yc, y64, mag) < TOL


def test_reference_dense_overlay_packing_and_emulator():
    rng = np.random.default_rng(5)
    A = (rng.random((1000, 20000), dtype=np.float32) - 0.5)
    x = rng.random(20000, dtype=np.float32)
    cin = rng.random(1000, dtype=np.float32)
    h = oracle.RefPack(24, 1, 1, 2, 5, True, False, True).prepare_dense(A)
    info = h.info()
    assert info["col_tiles"] == 3 and info["row_tiles"] == 1
    assert info["run_length"] == info["rows_per_pe"] * info["padded_cols"] // 2
    y = h.emulate(x, cin, 1.0, 1.0)
    ref = A.astype(np.float64) @ x.astype(np.float64) + cin
    mag = np.abs(A.astype(np.float64)) @ np.abs(x.astype(np.float64)) + np.abs(cin)
    assert bwd_err(y, ref, mag) < TOL
    yn = oracle.naive_gemv(A, x, cin, 1.0, 1.0)
    assert bwd_err(yn, ref, mag) < TOL
    with pytest.raises(AssertionError):
        oracle.RefPack(16, 1, 4, 2, 5, False, False, True).prepare_dense(A[:10, :10])   # no dense overlay
