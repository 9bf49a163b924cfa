"""Drop-in for the reference's pybind11 module ``pyhispmv`` (pyhispmv/src/pyhispmv_bindings.cpp:3-39):
``import pyhispmv; pyhispmv.FpgaHandle(...)`` / ``from pyhispmv import FpgaHandle`` keep working
(apps/general_test.py:2,22; apps/model_test.py:6), backed by the gfx950 HIP library."""
from hispmv_amd.fpga_handle import FpgaHandle

__all__ = ["FpgaHandle"]
__doc__ = "Python binding for the MI355X-native SpMV kernel (FpgaHandle-compatible)"
