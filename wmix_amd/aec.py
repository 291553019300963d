"""Host-side mirror of wmix's AEC wrapper (src/webrtc.h:40-45) for batches of near-end streams
that share one far-end reference.  All arithmetic happens in wmix_amd/csrc/aec.hip."""
import ctypes as C

import numpy as np
import torch

from .lifetime import Lifetime
from ._lib import check, lib, WmxError


class AecBatch(Lifetime):
    _mod = "aec"

    def __init__(self, n_streams, chn, freq, interval_ms=10, stream_far=None, n_cohorts=None):
        """stream_far: optional int array [n_streams], the far-end (0 .. n_far-1) each stream is cancelled against; the far
        tensors then carry a leading far-end dimension: [n_far, n_packets, pkt].
        n_cohorts: that many control cohorts instead (every stream starts in cohort 0; see run_cohorts)."""
        self._h = C.c_void_p()
        self.n_far = 1
        if n_cohorts is not None:
            assert stream_far is None
            self.n_far = int(n_cohorts)
            rc = lib().wmx_aec_create_groups(C.byref(self._h), n_streams, chn, freq, interval_ms, self.n_far, None)
        elif stream_far is None:
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
        self.n_cohorts = self.n_far
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

    def run_cohorts(self, far, near, delays, cohort_on=None, out=None, mode=3):
        """aec_process2 with one reported delay per cohort: far int16 CUDA [n_packets, pkt] (every cohort hears it, blocked
        from its own start), near [n_streams, n_packets, pkt]; delays / cohort_on: per-cohort sequences.  Returns
        (rc, per-cohort codes)."""
        assert far.is_cuda and far.dtype == torch.int16 and far.dim() == 2 and far.shape[1] == self.pkt and far.stride(1) == 1
        assert near.is_cuda and near.dtype == torch.int16 and near.dim() == 3 and near.stride(2) == 1 and near.shape[0] == self.n_streams
        out = near if out is None else out
        d = np.ascontiguousarray(delays, dtype=np.int32)
        assert d.shape == (self.n_far,)
        on = None if cohort_on is None else np.ascontiguousarray(cohort_on, dtype=np.uint8)
        codes = np.zeros(self.n_far, np.int32)
        rc = lib().wmx_aec_run_cohorts(self._h, mode, far.data_ptr(), far.stride(0), 0, near.data_ptr(), out.data_ptr(), near.shape[1],
                                       near.stride(0), near.stride(1), d.ctypes.data, None if on is None else on.ctypes.data,
                                       codes.ctypes.data, torch.cuda.current_stream().cuda_stream)
        if rc not in (0, -1):
            check(rc, "wmx_aec_run_cohorts")
        return rc, codes

    def set_timing(self, on):
        check(lib().wmx_aec_set_timing(self._h, 1 if on else 0), "wmx_aec_set_timing")

    def timing(self):
        """(launches, far-kernel ms, near-kernel ms) summed since the last call; waits for the last launch."""
        n, f, r = C.c_int(0), C.c_double(0), C.c_double(0)
        check(lib().wmx_aec_timing(self._h, C.byref(n), C.byref(f), C.byref(r)), "wmx_aec_timing")
        return n.value, f.value, r.value

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
