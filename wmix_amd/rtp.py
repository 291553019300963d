"""Host-side mirror of the RTP / G.711 packet edge (src/rtp.h, src/wmixTask.c:1019-1351) over torch device tensors.
All arithmetic happens in wmix_amd/csrc/rtp.hip."""
import ctypes as C

import torch

from ._lib import check, lib

LAW = {"a": 0, "u": 1, 0: 0, 1: 1}


class RtpSenders:
    """n streams of wmix_thread_rtp_send_pcma state (sequence number, timestamp)."""

    def __init__(self, n_streams, law="a"):
        self._h = C.c_void_p()
        check(lib().wmx_rtp_create(C.byref(self._h), n_streams, LAW[law]), "wmx_rtp_create")
        self.n = n_streams

    def egress(self, pcm, in_chn, in_freq, out_chn, out_freq, packets=None):
        """pcm int16 CUDA [n_streams, samples] -> uint8 CUDA [n_streams, packet_bytes] datagrams (header + codes)."""
        assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.dim() == 2 and pcm.shape[0] == self.n and pcm.stride(1) == 1
        if packets is None:
            packets = torch.zeros((self.n, 12 + pcm.shape[1] * max(1, (out_chn * out_freq + in_chn * in_freq - 1) // (in_chn * in_freq))),
                                  dtype=torch.uint8, device=pcm.device)
        size = C.c_uint32(0)
        check(lib().wmx_rtp_egress(self._h, in_chn, in_freq, pcm.data_ptr(), pcm.shape[1] * 2, pcm.stride(0), out_chn, out_freq,
                                   packets.data_ptr(), packets.stride(0), C.byref(size), torch.cuda.current_stream().cuda_stream),
              "wmx_rtp_egress")
        return packets[:, : size.value]

    def state(self, stream=0):
        s, t = C.c_uint16(0), C.c_uint32(0)
        check(lib().wmx_rtp_export(self._h, stream, C.byref(s), C.byref(t)), "wmx_rtp_export")
        return s.value, t.value

    def close(self):
        if self._h:
            lib().wmx_rtp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ingest(packets):
    """packets uint8 CUDA [n, >= 172] -> (pcm int16 [n, 160], pcm_bytes int32 [n], seq_raw int16 [n])"""
    assert packets.is_cuda and packets.dtype == torch.uint8 and packets.dim() == 2 and packets.stride(1) == 1
    n = packets.shape[0]
    pcm = torch.zeros((n, 160), dtype=torch.int16, device=packets.device)
    nbytes = torch.zeros(n, dtype=torch.int32, device=packets.device)
    seq = torch.zeros(n, dtype=torch.int16, device=packets.device)
    check(lib().wmx_rtp_ingest(n, packets.data_ptr(), packets.stride(0), pcm.data_ptr(), pcm.stride(0), nbytes.data_ptr(), seq.data_ptr(),
                               torch.cuda.current_stream().cuda_stream), "wmx_rtp_ingest")
    return pcm, nbytes, seq
