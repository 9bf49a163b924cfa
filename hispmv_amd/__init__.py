"""hispmv_amd -- MI355X (gfx950) native SpMV hot path of mfkiwl/HiSpMV behind the reference's
``pyhispmv.FpgaHandle`` interface.  The compute lives in ``lib/libhispmv.so`` (C ABI in
``include/hispmv.h``, hand-written HIP kernels in ``csrc/``); this package holds the ctypes
loader and the host-side mirror of the reference's Python-visible class."""
from .fpga_handle import FpgaHandle

__all__ = ["FpgaHandle"]
__version__ = "0.1.0"
