"""wmix_amd -- MI355X (gfx950) implementation of wmix's per-frame DSP hot path.

The product is libwmix_amd.so (hand-written HIP kernels behind a C ABI, see
include/wmix_amd.h and include/wmix_compat.h).  This Python package is only the
host-side mirror used by bench.py and the tests: it loads the library with
ctypes and passes torch device pointers through.  Nothing here computes audio
on the CPU; if the library is missing the import of `wmix_amd._lib.lib()` raises.
"""
__version__ = "0.1.0"
