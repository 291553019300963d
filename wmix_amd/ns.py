"""Host-side mirror of wmix's NS wrapper (src/webrtc.h:47-51) for batches of streams.

`NsBatch(n_streams, chn, freq)` == n_streams x ns_init(chn, freq, NULL);
`process(pcm)` == ns_process(fp, frame, frameOut, frameNum) on every stream, with the
PCM resident on the GPU.  All arithmetic happens in wmix_amd/csrc/ns.hip.
"""
import ctypes as C

import numpy as np
import torch

from .lifetime import Lifetime
from ._lib import check, lib


class NsBatch(Lifetime):
    _mod = "ns"

    def __init__(self, n_streams, chn, freq):
        self._h = C.c_void_p()
        L = lib()
        rc = L.wmx_ns_create(C.byref(self._h), n_streams, chn, freq)
        if rc != 0:
            self._h = None
            check(rc, "wmx_ns_create")  # raises; ns_init returns NULL for the same arguments
        self.n_streams, self.chn, self.freq = n_streams, chn, freq
        self.pkt = L.wmx_ns_packet_samples(self._h)  # int16 elements per 10 ms packet

    def process(self, pcm, out=None):
        """pcm: int16 CUDA tensor [n_streams, n_packets, pkt] (stream-major) or
        [n_packets, n_streams, pkt] with layout='packet' -- see process_strided."""
        assert pcm.dim() == 3 and pcm.shape[0] == self.n_streams and pcm.shape[2] == self.pkt
        return self.process_strided(pcm, pcm.shape[1], pcm.stride(0), pcm.stride(1), out)

    def process_packet_major(self, pcm, out=None):
        """pcm: [n_packets, n_streams, pkt] (one 10 ms step of all streams is contiguous)."""
        assert pcm.dim() == 3 and pcm.shape[1] == self.n_streams and pcm.shape[2] == self.pkt
        return self.process_strided(pcm, pcm.shape[0], pcm.stride(1), pcm.stride(0), out)

    def process_strided(self, pcm, n_packets, stream_stride, packet_stride, out=None):
        assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.stride(-1) == 1
        if out is None:
            out = pcm  # in place, like the daemon (src/wmix.c:622-626)
        assert out.is_cuda and out.dtype == torch.int16 and out.stride() == pcm.stride()
        check(lib().wmx_ns_process(self._h, pcm.data_ptr(), out.data_ptr(), n_packets, stream_stride, packet_stride,
                                   torch.cuda.current_stream().cuda_stream), "wmx_ns_process")
        return out

    def export_state(self, stream_index):
        L = lib()
        words = np.zeros(L.wmx_ns_state_words(self._h), np.float32)
        hist = np.zeros(3000, np.uint16)
        check(L.wmx_ns_export_state(self._h, stream_index, words.ctypes.data, hist.ctypes.data), "wmx_ns_export_state")
        return words, hist

    def close(self):
        if self._h:
            lib().wmx_ns_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
