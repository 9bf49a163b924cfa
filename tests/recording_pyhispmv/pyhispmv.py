"""A RECORDING stand-in for the `pyhispmv` module (test infrastructure; never on the product path, never shipped to
the GPU box as the thing under test): the same class and method names as the reference's binding
(pyhispmv/src/pyhispmv_bindings.cpp:3-39), numpy / scipy arithmetic in fp64, and a log of every call -- method, argument
shapes and dtypes, scalars -- written to $HISPMV_RECORD_OUT at exit.  tests/golden/make_apps_calls.py runs the reference's
apps/general_test.py and apps/model_test.py UNCHANGED against it in the build container to capture the call sequence and
the scripts' printed verdicts; the GPU test then replays the same sequence through the real module."""
import atexit
import json
import os

import numpy as np
import scipy.sparse as sp

_LOG = []


def _desc(a):
    a = np.asarray(a)
    return {"shape": list(a.shape), "dtype": str(a.dtype), "contiguous": bool(a.flags["C_CONTIGUOUS"])}


@atexit.register
def _dump():
    out = os.environ.get("HISPMV_RECORD_OUT")
    if out:
        with open(out, "w") as f:
            json.dump(_LOG, f)


class FpgaHandle:
    def __init__(self, xclbin_path, device_id, num_ch_A, num_ch_B, num_ch_C, urams_per_pe, fp_acc_latency, dense_overlay,
                 pre_accumulator, row_dist_net):
        _LOG.append({"call": "FpgaHandle", "xclbin_basename": os.path.basename(str(xclbin_path)), "device_id": int(device_id),
                     "hw": [int(num_ch_A), int(num_ch_B), int(num_ch_C), int(urams_per_pe), int(fp_acc_latency), bool(dense_overlay),
                            bool(pre_accumulator), bool(row_dist_net)]})
        self.mats, self.loaded, self.sel = [], False, None

    def create_dense_handle(self, flattened_dense_values, rows, cols):
        _LOG.append({"call": "create_dense_handle", "flattened_dense_values": _desc(flattened_dense_values), "rows": int(rows), "cols": int(cols)})
        self.mats.append(np.asarray(flattened_dense_values, np.float32).reshape(rows, cols))
        return len(self.mats) - 1

    def create_sparse_handle(self, coo_rows, coo_cols, coo_values, rows, cols):
        _LOG.append({"call": "create_sparse_handle", "coo_rows": _desc(coo_rows), "coo_cols": _desc(coo_cols), "coo_values": _desc(coo_values),
                     "rows": int(rows), "cols": int(cols)})
        self.mats.append(sp.coo_matrix((np.asarray(coo_values, np.float64), (np.asarray(coo_rows), np.asarray(coo_cols))), shape=(rows, cols)).tocsr())
        return len(self.mats) - 1

    def load_matrices(self):
        _LOG.append({"call": "load_matrices", "handles": len(self.mats)})
        self.loaded = True

    def select_matrix(self, matrix_idx):
        _LOG.append({"call": "select_matrix", "matrix_idx": int(matrix_idx)})
        self.sel = int(matrix_idx)

    def run_kernel(self, x, bias, y, alpha, beta):
        _LOG.append({"call": "run_kernel", "x": _desc(x), "bias": _desc(bias), "y": _desc(y), "alpha": float(alpha), "beta": float(beta)})
        assert self.loaded and self.sel is not None
        A = self.mats[self.sel]
        y[...] = (alpha * (A @ np.asarray(x, np.float64)) + beta * np.asarray(bias, np.float64)).astype(np.float32)

    def linear(self, matrix_idx, x, bias):
        A = self.mats[int(matrix_idx)]
        rows, cols = A.shape
        x = np.asarray(x)
        n = x.size // cols
        if not _LOG or _LOG[-1].get("call") != "linear" or _LOG[-1]["matrix_idx"] != int(matrix_idx):
            _LOG.append({"call": "linear", "matrix_idx": int(matrix_idx), "x": _desc(x), "bias": _desc(bias), "num_vecs": int(n), "times": 0})
        _LOG[-1]["times"] += 1
        X = x.reshape(n, cols).astype(np.float64)
        return ((A @ X.T).T + np.asarray(bias, np.float64)).astype(np.float32).reshape(-1)
