"""The packet edge end to end: what wmix_thread_rtp_recv_pcma / the record chain /
wmix_thread_rtp_send_pcma do for ONE stream per 20 ms (src/wmixTask.c:1278-1316, src/wmix.c:613-709,
src/wmixTask.c:1124-1143), for a batch of streams with only the 172-byte datagrams crossing PCIe:

    RTP/PCMA datagram -> header + A-law decode (rtp.hip) -> NS -> AEC -> AGC -> VAD (two 10 ms packets each, 8 kHz mono,
    in place) -> zoom 1x8000 -> A-law encode -> RTP header with the stream's running seq / timestamp (rtp.hip)

This file is a ctypes mirror: the sequencing -- pinned slots, the copy-in and copy-out HIP streams, the events, the three launches
per step -- lives in the library (wmix_amd/csrc/pipe.hip, wmx_pipe_*), where a C host finds it too (examples/host_rtp_pipe.c; its --pcm mode drives wmx_pipe_create_pcm).
`RtpChain.step()` works on datagrams already resident in HBM; `StreamingPipe` drives the slots: H2D of step k + 1 and D2H of
step k - 1 overlap the compute of step k.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib
from .chain import AEC, AGC, NS, VAD

DATAGRAM = 172  # 12-byte RTP header + 160 G.711 codes (20 ms at 8 kHz), src/rtp.h:33, src/rtp.c:86-95
FREQ, PKT = 8000, 80


class RtpChain:
    def __init__(self, n_streams, dev, agc_value=5, slots=3):
        self.n, self.dev = n_streams, dev
        self._h = C.c_void_p()
        check(lib().wmx_pipe_create(C.byref(self._h), n_streams, slots, 0, agc_value, NS | AEC | AGC | VAD), "wmx_pipe_create")
        self.slots = slots
        self.row_shape, self.row_dtype, self.far_shape = (n_streams, DATAGRAM), np.uint8, (2, PKT)

    def step(self, packets_in, far, packets_out):
        """packets_in / packets_out: uint8 CUDA [n_streams, 172]; far: int16 CUDA [2, 80], the shared far-end of these 20 ms."""
        assert packets_in.is_cuda and packets_out.is_cuda and packets_in.stride(1) == 1 and packets_out.stride(1) == 1
        assert far.is_cuda and far.dtype == torch.int16 and far.is_contiguous() and far.numel() == 2 * PKT
        check(lib().wmx_pipe_step_resident(self._h, packets_in.data_ptr(), packets_in.stride(0), far.data_ptr(), packets_out.data_ptr(),
                                           packets_out.stride(0), torch.cuda.current_stream().cuda_stream), "wmx_pipe_step_resident")
        return packets_out

    def close(self):
        if self._h:
            lib().wmx_pipe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PcmChain(RtpChain):
    """wmx_pipe_create_pcm: the heartbeat over PCM packages in host memory (src/wmix.c:609-709); rows int16 [n_streams, package],
    far-end int16 [packets, 10 ms packet]."""

    def __init__(self, n_streams, dev, chn=1, freq=16000, interval_ms=10, agc_value=5, stages=NS | AEC | AGC | VAD, slots=3):
        self.n, self.dev = n_streams, dev
        self._h = C.c_void_p()
        check(lib().wmx_pipe_create_pcm(C.byref(self._h), n_streams, slots, chn, freq, interval_ms, agc_value, stages), "wmx_pipe_create_pcm")
        self.slots = slots
        self.pkt10, self.ppc = freq // 100 * chn, interval_ms // 10
        assert lib().wmx_pipe_datagram_bytes(self._h) == self.pkt10 * self.ppc * 2
        self.row_shape, self.row_dtype, self.far_shape = (n_streams, self.pkt10 * self.ppc), np.int16, (self.ppc, self.pkt10)

    def step(self, pcm, far):
        """in place on packages already resident in HBM: pcm int16 CUDA [n_streams, package], far int16 CUDA [packets, 10 ms packet]"""
        assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.stride(1) == 1 and far.is_cuda and far.is_contiguous()
        check(lib().wmx_pipe_step_resident(self._h, pcm.data_ptr(), pcm.stride(0) * 2, far.data_ptr(), pcm.data_ptr(), pcm.stride(0) * 2,
                                           torch.cuda.current_stream().cuda_stream), "wmx_pipe_step_resident")
        return pcm


def _host_rows(ptr, shape, dtype):
    """numpy view of a pinned host buffer the library owns"""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    buf = (C.c_uint8 * n).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class StreamingPipe:
    """Host-resident datagrams in, host-resident datagrams out, copies overlapped with compute (wmx_pipe_submit / _wait)."""

    def __init__(self, chain):
        self.c = chain
        self.SLOTS = chain.slots
        L = lib()
        self.h_in = [_host_rows(L.wmx_pipe_in(chain._h, s), chain.row_shape, chain.row_dtype) for s in range(self.SLOTS)]
        self.h_out = [_host_rows(L.wmx_pipe_out(chain._h, s), chain.row_shape, chain.row_dtype) for s in range(self.SLOTS)]
        self.h_far = [_host_rows(L.wmx_pipe_far(chain._h, s), chain.far_shape, np.int16) for s in range(self.SLOTS)]

    def submit(self, far=None):
        """Process the datagrams the caller has placed in h_in[slot] (slots are taken round robin); far: int16 CUDA [2, 80], or
        None = the samples in h_far[slot].  Returns the slot; its h_out rows are valid after wait(slot)."""
        slot = C.c_int(-1)
        check(lib().wmx_pipe_submit(self.c._h, None if far is None else far.data_ptr(), C.byref(slot),
                                    torch.cuda.current_stream().cuda_stream), "wmx_pipe_submit")
        return slot.value

    def wait(self, slot=-1):
        check(lib().wmx_pipe_wait(self.c._h, slot), "wmx_pipe_wait")

    def drain(self):
        self.wait(-1)
