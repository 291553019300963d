"""Host-side mirror of wmix's VAD wrapper (src/webrtc.h:32-36) for batches of streams.
All arithmetic happens in wmix_amd/csrc/vad.hip."""
import ctypes as C

import torch

from .lifetime import Lifetime
from ._lib import check, lib


class VadBatch(Lifetime):
    _mod = "vad"

    def __init__(self, n_streams, chn, freq, interval_ms=10):
        self._h = C.c_void_p()
        rc = lib().wmx_vad_create(C.byref(self._h), n_streams, chn, freq, interval_ms)
        if rc != 0:
            self._h = None
            check(rc, "wmx_vad_create")
        self.n_streams, self.chn, self.freq = n_streams, chn, freq
        self.pkt = lib().wmx_vad_packet_samples(self._h)

    def process(self, pcm, packets_per_call=1):
        """pcm int16 CUDA [n_streams, n_calls, packets_per_call*pkt], modified in place."""
        assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.dim() == 3 and pcm.stride(2) == 1
        assert pcm.shape[0] == self.n_streams and pcm.shape[2] == packets_per_call * self.pkt
        check(lib().wmx_vad_process(self._h, pcm.data_ptr(), packets_per_call, pcm.shape[1], pcm.stride(0), pcm.stride(1),
                                    torch.cuda.current_stream().cuda_stream), "wmx_vad_process")
        return pcm

    def process_packet_major(self, pcm, packets_per_call=1):
        """pcm int16 CUDA [n_calls, n_streams, packets_per_call*pkt], modified in place."""
        assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.dim() == 3 and pcm.stride(2) == 1
        assert pcm.shape[1] == self.n_streams and pcm.shape[2] == packets_per_call * self.pkt
        check(lib().wmx_vad_process(self._h, pcm.data_ptr(), packets_per_call, pcm.shape[0], pcm.stride(1), pcm.stride(0),
                                    torch.cuda.current_stream().cuda_stream), "wmx_vad_process")
        return pcm

    def close(self):
        if self._h:
            lib().wmx_vad_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
