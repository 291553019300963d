"""Host-side mirror of wmix's AEC wrapper (src/webrtc.h:40-45) for batches of near-end streams
that share one far-end reference.  All arithmetic happens in wmix_amd/csrc/aec.hip."""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib, WmxError


class AecBatch:
    def __init__(self, n_streams, chn, freq, interval_ms=10, stream_far=None):
        """stream_far: optional int array [n_streams], the far-end (0 .. n_far-1) each stream is cancelled against; the far
        tensors then carry a leading far-end dimension: [n_far, n_packets, pkt]."""
        self._h = C.c_void_p()
        self.n_far = 1
        if stream_far is None:
            rc = lib().wmx_aec_create(C.byref(self._h), n_streams, chn, freq, interval_ms)
        else:
            m = np.ascontiguousarray(stream_far, dtype=np.int32)
            assert m.shape == (n_streams,)
            self.n_far = int(m.max()) + 1
            rc = lib().wmx_aec_create_groups(C.byref(self._h), n_streams, chn, freq, interval_ms, self.n_far, m.ctypes.data)
        if rc != 0:
            self._h = None
            check(rc, "wmx_aec_create")
        self.n_streams, self.chn, self.freq = n_streams, chn, freq
        self.pkt = lib().wmx_aec_packet_samples(self._h)

    def _run(self, mode, far, near, out, n_packets, stream_stride, packet_stride, delay_ms):
        fp = far.data_ptr() if far is not None else None
        gs = 0
        if far is not None and far.dim() == 3:  # [n_far, n_packets, pkt]
            assert far.shape[0] == self.n_far
            gs, far = far.stride(0), far[0]
        fs = far.stride(0) if far is not None else 0
        rc = lib().wmx_aec_run_groups(self._h, mode, fp, fs, gs, near.data_ptr() if near is not None else None,
                                      out.data_ptr() if out is not None else None, n_packets, stream_stride, packet_stride, delay_ms,
                                      torch.cuda.current_stream().cuda_stream)
        if rc not in (0, -1):
            check(rc, "wmx_aec_run")
        return rc

    def process2(self, far, near, out=None, delay_ms=0):
        """aec_process2: far int16 CUDA [n_packets, pkt] (shared) or [n_far, n_packets, pkt], near [n_streams, n_packets, pkt]."""
        assert far.is_cuda and far.dtype == torch.int16 and far.dim() in (2, 3) and far.shape[-1] == self.pkt and far.stride(-1) == 1
        assert near.is_cuda and near.dtype == torch.int16 and near.dim() == 3 and near.stride(2) == 1
        assert near.shape == (self.n_streams, far.shape[-2], self.pkt)
        out = near if out is None else out
        assert out.stride() == near.stride()
        return self._run(3, far, near, out, far.shape[-2], near.stride(0), near.stride(1), delay_ms), out

    def process2_packet_major(self, far, near, out=None, delay_ms=0):
        """near int16 CUDA [n_packets, n_streams, pkt] (one 10 ms step of all streams contiguous)."""
        assert far.is_cuda and far.dtype == torch.int16 and far.dim() == 2 and far.shape[1] == self.pkt and far.stride(1) == 1
        assert near.is_cuda and near.dtype == torch.int16 and near.dim() == 3 and near.stride(2) == 1
        assert near.shape == (far.shape[0], self.n_streams, self.pkt)
        out = near if out is None else out
        assert out.stride() == near.stride()
        return self._run(3, far, near, out, far.shape[0], near.stride(1), near.stride(0), delay_ms), out

    def set_frame_far(self, far):
        assert far.is_cuda and far.dtype == torch.int16 and far.dim() == 2 and far.shape[1] == self.pkt
        return self._run(1, far, None, None, far.shape[0], 0, 0, 0)

    def process(self, near, out=None, delay_ms=0):
        assert near.is_cuda and near.dtype == torch.int16 and near.dim() == 3 and near.shape[0] == self.n_streams
        out = near if out is None else out
        return self._run(2, None, near, out, near.shape[1], near.stride(0), near.stride(1), delay_ms), out

    def export_state(self, stream_index):
        w = np.zeros(lib().wmx_aec_state_words(self._h), np.float32)
        check(lib().wmx_aec_export_state(self._h, stream_index, w.ctypes.data), "wmx_aec_export_state")
        return w

    def close(self):
        if self._h:
            lib().wmx_aec_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
