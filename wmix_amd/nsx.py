"""Host-side mirror of wmix's NS wrapper built with MAKE_WEBRTC_NSX (src/webrtc.c:512-521) for batches of streams:
the fixed-point noise suppressor.  Same call shapes as wmix_amd/ns.py; all arithmetic is in wmix_amd/csrc/nsx.hip."""
import ctypes as C

import torch

from .lifetime import Lifetime
from ._lib import check, lib


class NsxBatch(Lifetime):
    _mod = "nsx"

    def __init__(self, n_streams, chn, freq):
        self._h = C.c_void_p()
        L = lib()
        rc = L.wmx_nsx_create(C.byref(self._h), n_streams, chn, freq)
        if rc != 0:
            self._h = None
            check(rc, "wmx_nsx_create")
        self.n_streams, self.chn, self.freq = n_streams, chn, freq
        self.pkt = L.wmx_nsx_packet_samples(self._h)
        self.state_bytes = L.wmx_nsx_state_bytes(self._h)

    def process(self, pcm, out=None):
        """pcm: int16 CUDA tensor [n_streams, n_packets, pkt]; in place unless `out` is given."""
        assert pcm.dim() == 3 and pcm.shape[0] == self.n_streams and pcm.shape[2] == self.pkt
        return self.process_strided(pcm, pcm.shape[1], pcm.stride(0), pcm.stride(1), out)

    def process_packet_major(self, pcm, out=None):
        """pcm: [n_packets, n_streams, pkt] (one 10 ms step of all streams is contiguous)."""
        assert pcm.dim() == 3 and pcm.shape[1] == self.n_streams and pcm.shape[2] == self.pkt
        return self.process_strided(pcm, pcm.shape[0], pcm.stride(1), pcm.stride(0), out)

    def process_strided(self, pcm, n_packets, stream_stride, packet_stride, out=None):
        assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.stride(-1) == 1
        if out is None:
            out = pcm
        assert out.is_cuda and out.dtype == torch.int16 and out.stride() == pcm.stride()
        check(lib().wmx_nsx_process(self._h, pcm.data_ptr(), out.data_ptr(), n_packets, stream_stride, packet_stride,
                                    torch.cuda.current_stream().cuda_stream), "wmx_nsx_process")
        return out

    def close(self):
        if self._h:
            lib().wmx_nsx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
