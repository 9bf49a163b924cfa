"""Stand-in for the ``sparse_dot_mkl`` package the reference's apps import (apps/model.py:5,42:
``dot_product_mkl(csr_weight, x.T)``), which is not installable here (no package index).  Same call, same result
type; the product is computed by scipy's CSR kernels instead of MKL's -- it is the CPU comparison path of the apps,
never the MI355X path (that is pyhispmv.FpgaHandle.linear)."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

__all__ = ["dot_product_mkl"]


def dot_product_mkl(matrix_a, matrix_b, cast: bool = False, copy: bool = True, reorder_output: bool = False,
                    dense: bool = False, debug: bool = False, out=None, out_scalar=None):
    """matrix_a @ matrix_b for scipy sparse and/or numpy operands (the subset of the real package's signature the
    apps use: sparse @ dense -> dense ndarray, sparse @ sparse -> sparse unless dense=True)."""
    a_sparse, b_sparse = sp.issparse(matrix_a), sp.issparse(matrix_b)
    if not a_sparse and not b_sparse:
        res = np.asarray(matrix_a) @ np.asarray(matrix_b)
    else:
        res = matrix_a @ matrix_b
        if sp.issparse(res) and dense:
            res = res.toarray()
        elif not sp.issparse(res):
            res = np.asarray(res)
    if out is not None:
        if out_scalar is not None:
            out *= out_scalar
            out += res
        else:
            out[...] = res
        return out
    return res
