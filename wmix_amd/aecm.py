"""Host-side mirror of wmix's AEC wrapper built with the reference's AECM switch (src/webrtc.c:168-191) for batches of
near-end streams that share one far-end: the fixed-point echo canceller.  Same call shapes as wmix_amd/aec.py; all
arithmetic is in wmix_amd/csrc/aecm.hip."""
import ctypes as C

import torch

from .lifetime import Lifetime
from ._lib import check, lib


class AecmBatch(Lifetime):
    _mod = "aecm"

    def __init__(self, n_streams, chn, freq, interval_ms=10, n_cohorts=1):
        self._h = C.c_void_p()
        self.n_cohorts = int(n_cohorts)
        rc = lib().wmx_aecm_create_cohorts(C.byref(self._h), n_streams, chn, freq, interval_ms, self.n_cohorts)
        if rc != 0:
            self._h = None
            check(rc, "wmx_aecm_create")
        self.n_streams, self.chn, self.freq = n_streams, chn, freq
        self.pkt = lib().wmx_aecm_packet_samples(self._h)
        self.state_bytes = lib().wmx_aecm_state_bytes(self._h)

    def _run(self, mode, far, near, out, n_packets, stream_stride, packet_stride, delay_ms):
        fp = far.data_ptr() if far is not None else None
        fs = far.stride(0) if far is not None else 0
        rc = lib().wmx_aecm_run(self._h, mode, fp, fs, near.data_ptr() if near is not None else None,
                                out.data_ptr() if out is not None else None, n_packets, stream_stride, packet_stride, delay_ms,
                                torch.cuda.current_stream().cuda_stream)
        if rc not in (0, -1):
            check(rc, "wmx_aecm_run")
        return rc

    def run_cohorts(self, far, near, delays, cohort_on=None, out=None, mode=3):
        """As AecBatch.run_cohorts: one reported delay (and optional on/off byte) per cohort; returns (rc, per-cohort codes)."""
        import numpy as np
        assert far.is_cuda and far.dtype == torch.int16 and far.dim() == 2 and far.shape[1] == self.pkt and far.stride(1) == 1
        assert near.is_cuda and near.dtype == torch.int16 and near.dim() == 3 and near.stride(2) == 1 and near.shape[0] == self.n_streams
        out = near if out is None else out
        d = np.ascontiguousarray(delays, dtype=np.int32)
        assert d.shape == (self.n_cohorts,)
        on = None if cohort_on is None else np.ascontiguousarray(cohort_on, dtype=np.uint8)
        codes = np.zeros(self.n_cohorts, np.int32)
        rc = lib().wmx_aecm_run_cohorts(self._h, mode, far.data_ptr(), far.stride(0), 0, near.data_ptr(), out.data_ptr(), near.shape[1],
                                        near.stride(0), near.stride(1), d.ctypes.data, None if on is None else on.ctypes.data,
                                        codes.ctypes.data, torch.cuda.current_stream().cuda_stream)
        if rc not in (0, -1):
            check(rc, "wmx_aecm_run_cohorts")
        return rc, codes

    def process2(self, far, near, out=None, delay_ms=0):
        """aec_process2: far int16 CUDA [n_packets, pkt] (shared), near [n_streams, n_packets, pkt]."""
        assert far.is_cuda and far.dtype == torch.int16 and far.dim() == 2 and far.shape[1] == self.pkt and far.stride(1) == 1
        assert near.is_cuda and near.dtype == torch.int16 and near.dim() == 3 and near.stride(2) == 1
        assert near.shape == (self.n_streams, far.shape[0], self.pkt)
        out = near if out is None else out
        assert out.stride() == near.stride()
        return self._run(3, far, near, out, far.shape[0], near.stride(0), near.stride(1), delay_ms), out

    def process2_packet_major(self, far, near, out=None, delay_ms=0):
        """near int16 CUDA [n_packets, n_streams, pkt] (one step of all streams contiguous)."""
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

    def close(self):
        if self._h:
            lib().wmx_aecm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
