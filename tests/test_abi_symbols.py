"""The C-ABI library loads on a CPU-only box and exports every symbol include/hispmv.h declares;
argument errors and the missing-device error are reported through return codes (no compute here)."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / "include" / "hispmv.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hispmv_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from hispmv_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 30
    raw = C.CDLL(str(_lib.LIB_PATH))
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/hispmv.h but not exported"
    assert set(syms) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    assert _lib.lib.hispmv_version().decode().endswith("gfx950")


def test_create_argument_errors_mirror_the_reference():
    from hispmv_amd import _lib
    lib = _lib.lib
    ctx = C.c_void_p()
    # fpga_handle.cpp:51-52 negative device id, :70-71 empty xclbin path
    assert lib.hispmv_create(C.byref(ctx), b"a.xclbin", -1, 24, 1, 1, 2, 5, 1, 0, 1) == _lib.HISPMV_EINVAL
    assert b"non-negative" in lib.hispmv_last_error(None)
    assert lib.hispmv_create(C.byref(ctx), b"", 0, 24, 1, 1, 2, 5, 1, 0, 1) == _lib.HISPMV_EINVAL
    assert b"XCLBIN path is empty" in lib.hispmv_last_error(None)
    # spmv-helper.cpp:15: num_pes must be a multiple of the output-channel width
    assert lib.hispmv_create(C.byref(ctx), b"a", 0, 1, 1, 1, 2, 5, 1, 0, 1) == _lib.HISPMV_EINVAL
    assert not ctx.value


def test_device_entry_points_check_their_arguments_before_touching_the_device():
    """Argument errors of the device-pointer entry points come back as HISPMV_EINVAL without a launch (so they can be
    checked on a CPU-only box): NULL context, negative counts, missing tables."""
    from hispmv_amd import _lib
    lib = _lib.lib
    assert lib.hispmv_spmv_device_batch(None, 0, None, None, None, None, 1.0, 1.0, None) == _lib.HISPMV_EINVAL
    assert lib.hispmv_spmv_device(None, 0, None, None, None, 1.0, 1.0, None) == _lib.HISPMV_EINVAL
    assert lib.hispmv_boundary_pack(None, None, None, None, -1, None) == _lib.HISPMV_EINVAL      # no context
    assert lib.hispmv_boundary_pack(None, None, None, None, 3, None) == _lib.HISPMV_EINVAL
    assert lib.hispmv_boundary_apply(None, None, None, None, 3, 2, None) == _lib.HISPMV_EINVAL
    assert lib.hispmv_boundary_apply(None, None, None, None, 0, 0, None) == _lib.HISPMV_EINVAL
    assert lib.hispmv_synchronize(None) == _lib.HISPMV_EINVAL
    out = (C.c_int64 * 4)()
    assert lib.hispmv_batch_call_info(None, out) == _lib.HISPMV_EINVAL                          # diagnostics of the batch entry point
    assert lib.hispmv_batch_graph_stats(None, out) == _lib.HISPMV_EINVAL
    # the host-only order of the step kernel's queue: bad arguments are refused, nothing is written
    cls = (C.c_int32 * 2)(7, 7)
    cost = (C.c_double * 1)(1.0)
    assert lib.hispmv_prep_step_queue(cost, 1, cost, 1, 0, 0, cls, cls) == _lib.HISPMV_EINVAL   # no workgroups
    assert lib.hispmv_prep_step_queue(cost, 1, cost, 1, 4, 9, cls, cls) == _lib.HISPMV_EINVAL   # unknown mode
    assert lib.hispmv_prep_step_queue(None, 1, cost, 1, 4, 0, cls, cls) == _lib.HISPMV_EINVAL   # missing table
    assert list(cls) == [7, 7]


def test_facade_fails_loudly_without_a_device():
    import os
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    import pyhispmv
    with pytest.raises(RuntimeError, match="no HIP device|device"):
        pyhispmv.FpgaHandle("x.xclbin", 0, 24, 1, 1, 2, 5, True, False, True)
    with pytest.raises(ValueError):
        pyhispmv.FpgaHandle("x.xclbin", -1, 24, 1, 1, 2, 5, True, False, True)


def test_product_does_not_touch_the_oracle():
    # the oracle is test infrastructure: nothing under hispmv_amd/ or pyhispmv/ may reference it
    for d in ("hispmv_amd", "pyhispmv", "include"):
        for f in (ROOT / d).rglob("*"):
            if f.suffix in {".py", ".cpp", ".hip", ".h"} and "build" not in f.parts:
                txt = f.read_text()
                assert "liboracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f
