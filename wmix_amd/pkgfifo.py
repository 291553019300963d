"""Host-side mirror of the daemon's AEC packet FIFOs (src/wmix.c:432-526) over torch device tensors.
All data movement happens in wmix_amd/csrc/pkgfifo.hip."""
import ctypes as C

import torch

from ._lib import check, lib


class PkgFifo:
    def __init__(self, n_streams, pkg_bytes=320, n_slots=22, interval_ms=20, frame_bytes=2):
        """defaults = the reference's 1 x 8000 Hz ring: WMIX_PKG_SIZE 320, AEC_FIFO_PKG_NUM 22, WMIX_INTERVAL_MS 20"""
        self._h = C.c_void_p()
        check(lib().wmx_pkgfifo_create(C.byref(self._h), n_streams, n_slots, pkg_bytes, interval_ms, frame_bytes), "wmx_pkgfifo_create")
        self.n, self.pkg = n_streams, pkg_bytes

    def add(self, pkgs):
        assert pkgs.is_cuda and pkgs.dtype == torch.uint8 and pkgs.shape == (self.n, self.pkg) and pkgs.stride(1) == 1
        check(lib().wmx_pkgfifo_add(self._h, pkgs.data_ptr(), pkgs.stride(0), torch.cuda.current_stream().cuda_stream), "wmx_pkgfifo_add")

    def get(self, delayms):
        out = torch.empty((self.n, self.pkg), dtype=torch.uint8, device="cuda")
        check(lib().wmx_pkgfifo_get(self._h, out.data_ptr(), out.stride(0), delayms, torch.cuda.current_stream().cuda_stream), "wmx_pkgfifo_get")
        return out

    def close(self):
        if self._h:
            lib().wmx_pkgfifo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
