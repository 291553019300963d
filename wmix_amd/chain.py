"""Host-side mirror of the daemon's record heartbeat (src/wmix.c:613-709) for a batch of streams: ns_process ->
aec_process2 -> agc_process -> vad_process on one buffer, ONE C call per tick (wmx_chain_process, wmix_amd/csrc/chain.hip)."""
import ctypes as C

import numpy as np
import torch

from .lifetime import Lifetime
from ._lib import check, lib

NS, AEC, AGC, VAD = 1, 2, 4, 8
NSX, AECM = 16, 32  # the NS / AEC stage is the reference's fixed-point build of it (WebRtcNsx_* / WebRtcAecm_*)


class ChainBatch(Lifetime):
    _mod = "chain"

    def __init__(self, n_streams, chn, freq, interval_ms=10, agc_value=5, stages=NS | AEC | AGC | VAD, n_cohorts=1, stream_cohort=None):
        """stream_cohort: optional int array [n_streams], the cohort (control plane + far-end) each stream belongs to from the start;
        `far` may then carry one far-end per cohort: [n10, n_cohorts, pkt] (wmx_chain_process_groups)."""
        self._h = C.c_void_p()
        m = None
        if stream_cohort is not None:
            m = np.ascontiguousarray(stream_cohort, dtype=np.int32)
            assert m.shape == (n_streams,)  # (the range is the library's to refuse: WMX_EINVAL)
        rc = lib().wmx_chain_create_groups(C.byref(self._h), n_streams, chn, freq, interval_ms, agc_value, stages, n_cohorts,
                                           None if m is None else m.ctypes.data)
        if rc != 0:
            self._h = None
            check(rc, "wmx_chain_create")
        self.n_streams, self.chn, self.freq, self.stages, self.n_cohorts = n_streams, chn, freq, stages, n_cohorts
        self.pkt = freq // 100 * chn  # int16 elements of a 10 ms packet

    def set_stages(self, stages, agc_value=-1):
        """webrtcEnable[] at run time (wmx_chain_set_stages): a stage that goes is released, one that comes on is made anew."""
        check(lib().wmx_chain_set_stages(self._h, int(stages), int(agc_value)), "wmx_chain_set_stages")
        self.stages = stages
        self.n_cohorts = lib().wmx_chain_cohorts(self._h)

    def _process(self, far, pcm, out, n10, stream_stride, packet_stride, delays, cohort_on):
        assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.stride(-1) == 1
        out = pcm if out is None else out
        assert out.stride() == pcm.stride()
        fp, fs, gs = None, 0, 0
        if far is not None and far.dim() == 3:  # one far-end per cohort: [n10, n_cohorts, pkt]
            assert far.is_cuda and far.dtype == torch.int16 and far.shape == (n10, self.n_cohorts, self.pkt) and far.stride(2) == 1
            fp, fs, gs = far.data_ptr(), far.stride(0), far.stride(1)
        elif far is not None:
            assert far.is_cuda and far.dtype == torch.int16 and far.dim() == 2 and far.shape == (n10, self.pkt) and far.stride(1) == 1
            fp, fs = far.data_ptr(), far.stride(0)
        d = None if delays is None else np.ascontiguousarray(delays, dtype=np.int32)
        on = None if cohort_on is None else np.ascontiguousarray(cohort_on, dtype=np.uint8)
        codes = np.zeros(self.n_cohorts, np.int32)
        rc = lib().wmx_chain_process_groups(self._h, fp, fs, gs, pcm.data_ptr(), out.data_ptr(), n10, stream_stride, packet_stride,
                                            None if d is None else d.ctypes.data, None if on is None else on.ctypes.data, codes.ctypes.data,
                                            torch.cuda.current_stream().cuda_stream)
        if rc not in (0, -1):
            check(rc, "wmx_chain_process")
        return rc, codes, out

    def process(self, far, pcm, out=None, delays=None, cohort_on=None):
        """One tick: far int16 CUDA [n10, pkt] (the shared far-end), pcm [n_streams, n10, pkt]; in place unless `out`."""
        assert pcm.dim() == 3 and pcm.shape[0] == self.n_streams and pcm.shape[2] == self.pkt
        return self._process(far, pcm, out, pcm.shape[1], pcm.stride(0), pcm.stride(1), delays, cohort_on)

    def process_packet_major(self, far, pcm, out=None, delays=None, cohort_on=None):
        """pcm [n10, n_streams, pkt]: one 10 ms step of all streams contiguous."""
        assert pcm.dim() == 3 and pcm.shape[1] == self.n_streams and pcm.shape[2] == self.pkt
        return self._process(far, pcm, out, pcm.shape[0], pcm.stride(1), pcm.stride(0), delays, cohort_on)

    def aec_handle(self):
        return lib().wmx_chain_aec(self._h)

    def set_aec_timing(self, on):
        """HIP events around the AEC's two kernels inside the library, on the launch stream (wmx_aec_set_timing)."""
        check(lib().wmx_aec_set_timing(self.aec_handle(), 1 if on else 0), "wmx_aec_set_timing")

    def aec_timing(self):
        """(launches, far-kernel ms, near-kernel ms) summed since the last call; waits for the last launch."""
        n, f, r = C.c_int(0), C.c_double(0), C.c_double(0)
        check(lib().wmx_aec_timing(self.aec_handle(), C.byref(n), C.byref(f), C.byref(r)), "wmx_aec_timing")
        return n.value, f.value, r.value

    def aec_host_ctl(self):
        """(launches, seconds) the AEC's host control planes took on this thread since the last call (wmx_aec_host_ctl)."""
        n, sec = C.c_long(0), C.c_double(0)
        check(lib().wmx_aec_host_ctl(self.aec_handle(), C.byref(n), C.byref(sec)), "wmx_aec_host_ctl")
        return n.value, sec.value

    def reset_streams_gain(self, idx, agc_value, cohort=None):
        """wmx_chain_reset_streams with agc_init's own `value` for the new handles (src/wmix.c:684)."""
        a = np.ascontiguousarray(idx, dtype=np.int32)
        check(lib().wmx_chain_reset_streams_gain(self._h, a.ctypes.data, a.size, -1 if cohort is None else int(cohort), int(agc_value),
                                                 torch.cuda.current_stream().cuda_stream), "wmx_chain_reset_streams_gain")

    def set_agc_gain_streams(self, idx, agc_value):
        """agc_addition(fp, value) for the listed streams of the running chain (src/wmix.c:1068-1070)."""
        a = np.ascontiguousarray(idx, dtype=np.int32)
        check(lib().wmx_chain_set_agc_gain_streams(self._h, a.ctypes.data, a.size, int(agc_value), torch.cuda.current_stream().cuda_stream),
              "wmx_chain_set_agc_gain_streams")

    def stage_calls_packet_major(self, far, pcm, out):
        """pcm / out [n10, n_streams, pkt] (see stage_calls)."""
        return self.stage_calls(far, pcm, out, pcm.shape[0], pcm.stride(1), pcm.stride(0))

    def stage_calls_stream_major(self, far, pcm, out):
        """pcm / out [n_streams, n10, pkt]: a stream's tick in one piece (what 20 ms VAD / 8 kHz AEC packets need)."""
        return self.stage_calls(far, pcm, out, pcm.shape[1], pcm.stride(0), pcm.stride(1))

    def stage_calls(self, far, pcm, out, n10, ss, ps):
        """The launches wmx_chain_process makes for a tick, one callable per stage, for callers that want an event between
        the stages (bench.py's per-stage breakdown).  Same handles, same state, same packet arithmetic as chain.hip."""
        L, st = lib(), torch.cuda.current_stream().cuda_stream
        total = n10 * self.pkt
        calls, src = [], pcm
        if self.stages & NSX:
            calls.append(("ns", lambda s=src: check(L.wmx_nsx_process(L.wmx_chain_nsx(self._h), s.data_ptr(), out.data_ptr(), n10, ss, ps, st))))
            src = out
        elif self.stages & NS:
            calls.append(("ns", lambda s=src: check(L.wmx_ns_process(L.wmx_chain_ns(self._h), s.data_ptr(), out.data_ptr(), n10, ss, ps, st))))
            src = out
        if self.stages & AECM:
            hm = L.wmx_chain_aecm(self._h)
            per = L.wmx_aecm_packet_samples(hm) // self.pkt
            calls.append(("aec", lambda s=src: check(L.wmx_aecm_run(hm, 3, far.data_ptr(), far.stride(0) * per, s.data_ptr(), out.data_ptr(),
                                                                    n10 // per, ss, ps * per, 0, st))))
            src = out
        elif self.stages & AEC:
            per = L.wmx_aec_packet_samples(self.aec_handle()) // self.pkt  # 10 ms packets per AEC packet (2 at 8 kHz, interval 20)
            calls.append(("aec", lambda s=src: check(L.wmx_aec_run(self.aec_handle(), 3, far.data_ptr(), far.stride(0) * per, s.data_ptr(),
                                                                   out.data_ptr(), n10 // per, ss, ps * per, 0, st))))
            src = out
        if self.stages & AGC:
            agc_pkg = L.wmx_agc_packet_samples(L.wmx_chain_agc(self._h))
            assert agc_pkg == self.pkt, "5 ms AGC packets (32 kHz): use wmx_chain_process"
            calls.append(("agc", lambda s=src: check(L.wmx_agc_process(L.wmx_chain_agc(self._h), s.data_ptr(), out.data_ptr(), n10, ss, ps, st))))
            src = out
        if self.stages & VAD:
            vad_pkg = L.wmx_vad_packet_samples(L.wmx_chain_vad(self._h))
            calls.append(("vad", lambda: check(L.wmx_vad_process(L.wmx_chain_vad(self._h), out.data_ptr(), total // vad_pkg, 1, ss,
                                                                 total if self.chn > 1 else vad_pkg, st))))
        return calls

    def close(self):
        if self._h:
            lib().wmx_chain_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
