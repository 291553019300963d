"""Host-side mirror of wmix's AGC wrapper (src/webrtc.h:55-60) for batches of streams.
All arithmetic happens in wmix_amd/csrc/agc.hip."""
import ctypes as C

import numpy as np
import torch

from .lifetime import Lifetime
from ._lib import check, lib


class AgcBatch(Lifetime):
    _mod = "agc"

    def __init__(self, n_streams, chn, freq, value, interval_ms=10):
        self._h = C.c_void_p()
        rc = lib().wmx_agc_create(C.byref(self._h), n_streams, chn, freq, interval_ms, value)
        if rc != 0:
            self._h = None
            check(rc, "wmx_agc_create")
        self.n_streams, self.chn, self.freq = n_streams, chn, freq
        self.pkt = lib().wmx_agc_packet_samples(self._h)

    def set_gain(self, value):
        check(lib().wmx_agc_set_gain(self._h, value), "wmx_agc_set_gain")

    def set_gain_streams(self, idx, value):
        """agc_addition(fp, value) for the listed streams (src/webrtc.c:824-839), ordered on the current HIP stream."""
        a = np.ascontiguousarray(idx, dtype=np.int32)
        check(lib().wmx_agc_set_gain_streams(self._h, a.ctypes.data, a.size, int(value), torch.cuda.current_stream().cuda_stream),
              "wmx_agc_set_gain_streams")

    def reset_streams_gain(self, idx, value):
        """agc_release + agc_init(chn, freq, intervalMs, value) for the listed streams."""
        a = np.ascontiguousarray(idx, dtype=np.int32)
        check(lib().wmx_agc_reset_streams_gain(self._h, a.ctypes.data, a.size, int(value), torch.cuda.current_stream().cuda_stream),
              "wmx_agc_reset_streams_gain")

    def stream_gain(self, stream_index):
        v = lib().wmx_agc_stream_gain(self._h, int(stream_index))
        if v < -10000:
            check(v, "wmx_agc_stream_gain")
        return v

    def gain_table(self):
        t = np.zeros(32, np.int32)
        check(lib().wmx_agc_gain_table(self._h, t.ctypes.data), "wmx_agc_gain_table")
        return t

    def _run(self, pcm, out, n_packets, stream_stride, packet_stride):
        assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.stride(-1) == 1
        if out is None:
            out = pcm
        assert out.stride() == pcm.stride() and out.dtype == torch.int16 and out.is_cuda
        check(lib().wmx_agc_process(self._h, pcm.data_ptr(), out.data_ptr(), n_packets, stream_stride, packet_stride,
                                    torch.cuda.current_stream().cuda_stream), "wmx_agc_process")
        return out

    def process(self, pcm, out=None):
        """pcm int16 CUDA [n_streams, n_packets, pkt]"""
        assert pcm.dim() == 3 and pcm.shape[0] == self.n_streams and pcm.shape[2] == self.pkt
        return self._run(pcm, out, pcm.shape[1], pcm.stride(0), pcm.stride(1))

    def process_packet_major(self, pcm, out=None):
        """pcm int16 CUDA [n_packets, n_streams, pkt]"""
        assert pcm.dim() == 3 and pcm.shape[1] == self.n_streams and pcm.shape[2] == self.pkt
        return self._run(pcm, out, pcm.shape[0], pcm.stride(1), pcm.stride(0))

    def close(self):
        if self._h:
            lib().wmx_agc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
