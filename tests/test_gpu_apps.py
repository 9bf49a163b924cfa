"""SURVEY.md 8(f)-1, the drop-in claim checked on both sides: examples/general_check.py and examples/model_check.py (this
repo's counterparts of the reference's apps/general_test.py:22-113 and apps/model_test.py:32-90 + apps/model.py:57-80 +
apps/fpga_layer_manager.py:15-80) run at the REFERENCE sizes on the MI355X through the real pyhispmv module, and the
sequence of FpgaHandle calls they make -- method order, argument shapes and dtypes, alpha/beta, the 100 `linear` calls per
layer -- must equal the sequence the reference's own scripts made when they ran UNCHANGED against the recording stand-in
in the build container (tests/golden/apps_calls.json, written by tests/golden/make_apps_calls.py; the reference's .py files
stay there).  Equal shapes of the seeded sparse layers (6 713 326 and 2 098 206 entries) also show that the examples draw
their random numbers in the reference's order.  Verdicts: the scripts' own np.allclose(rtol=1e-3) lines and the 1e-5 gate
(the examples exit non-zero otherwise)."""
import json
import runpy
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]
FIXTURE = json.loads((ROOT / "tests" / "golden" / "apps_calls.json").read_text())


def _desc(a):
    a = np.asarray(a)
    return {"shape": list(a.shape), "dtype": str(a.dtype), "contiguous": bool(a.flags["C_CONTIGUOUS"])}


def _recording(real_cls, log):
    """The real FpgaHandle behind a proxy that logs what tests/recording_pyhispmv logs."""
    class Proxy:
        def __init__(self, xclbin_path, device_id, *hw):
            log.append({"call": "FpgaHandle", "device_id": int(device_id), "hw": [int(v) if not isinstance(v, bool) else v for v in hw]})
            self._h = real_cls(xclbin_path, device_id, *hw)

        def create_dense_handle(self, flattened_dense_values, rows, cols):
            log.append({"call": "create_dense_handle", "flattened_dense_values": _desc(flattened_dense_values), "rows": int(rows), "cols": int(cols)})
            return self._h.create_dense_handle(flattened_dense_values, rows, cols)

        def create_sparse_handle(self, coo_rows, coo_cols, coo_values, rows, cols):
            log.append({"call": "create_sparse_handle", "coo_rows": _desc(coo_rows), "coo_cols": _desc(coo_cols), "coo_values": _desc(coo_values),
                        "rows": int(rows), "cols": int(cols)})
            return self._h.create_sparse_handle(coo_rows, coo_cols, coo_values, rows, cols)

        def load_matrices(self):
            log.append({"call": "load_matrices", "handles": self._h.num_matrices() if hasattr(self._h, "num_matrices") else None})
            return self._h.load_matrices()

        def select_matrix(self, matrix_idx):
            log.append({"call": "select_matrix", "matrix_idx": int(matrix_idx)})
            return self._h.select_matrix(matrix_idx)

        def run_kernel(self, x, bias, y, alpha, beta):
            log.append({"call": "run_kernel", "x": _desc(x), "bias": _desc(bias), "y": _desc(y), "alpha": float(alpha), "beta": float(beta)})
            return self._h.run_kernel(x, bias, y, alpha, beta)

        def linear(self, matrix_idx, x, bias):
            if not log or log[-1].get("call") != "linear" or log[-1]["matrix_idx"] != int(matrix_idx):
                log.append({"call": "linear", "matrix_idx": int(matrix_idx), "x": _desc(x), "bias": _desc(bias),
                            "num_vecs": int(np.asarray(x).size // self._h.matrix_info(matrix_idx)["cols"]), "times": 0})
            log[-1]["times"] += 1
            return self._h.linear(matrix_idx, x, bias)

        def __getattr__(self, name):
            return getattr(self._h, name)
    return Proxy


def _same_calls(got, want):
    """Call names in order; per call every field both sides recorded (the xclbin path and the handle count are environment)."""
    assert [c["call"] for c in got] == [c["call"] for c in want]
    for g, w in zip(got, want):
        for k, v in w.items():
            if k in ("xclbin_basename", "handles"):
                continue
            assert g.get(k) == v, (g["call"], k, g.get(k), v)


def _run(script, argv, monkeypatch, capsys):
    import pyhispmv
    log = []
    monkeypatch.setattr(pyhispmv, "FpgaHandle", _recording(pyhispmv.FpgaHandle, log))
    monkeypatch.setattr(sys, "argv", [str(script)] + argv)
    with pytest.raises(SystemExit) as ex:
        runpy.run_path(str(script), run_name="__main__")
    return log, capsys.readouterr().out, ex.value.code


def test_general_check_replays_the_reference_script_at_its_size(monkeypatch, capsys):
    log, out, code = _run(ROOT / "examples" / "general_check.py", [], monkeypatch, capsys)      # 50000 x 10000 dense + 1 M COO, np.random.seed(0)
    assert code == 0, out
    _same_calls(log, FIXTURE["general_test"]["calls"])
    got = [l.strip() for l in out.splitlines() if "result is" in l]
    assert got == FIXTURE["general_test"]["verdicts"] == ["Dense matrix result is correct!", "Sparse matrix result is correct!"]


def test_model_check_replays_the_reference_script_at_its_size(monkeypatch, capsys):
    log, out, code = _run(ROOT / "examples" / "model_check.py", [], monkeypatch, capsys)        # 4096 -> 8192 -> 8192 -> 1024, 100 calls per layer
    assert code == 0, out
    _same_calls(log, FIXTURE["model_test"]["calls"])
    assert "model check passed" in out
