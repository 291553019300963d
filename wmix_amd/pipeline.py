"""The packet edge end to end: what wmix_thread_rtp_recv_pcma / the record chain /
wmix_thread_rtp_send_pcma do for ONE stream per 20 ms (src/wmixTask.c:1278-1316, src/wmix.c:613-709,
src/wmixTask.c:1124-1143), for a batch of streams with only the 172-byte datagrams crossing PCIe:

    RTP/PCMA datagram -> header + A-law decode (rtp.hip) -> NS -> AEC -> AGC -> VAD (two 10 ms packets each, 8 kHz mono,
    in place) -> zoom 1x8000 -> A-law encode -> RTP header with the stream's running seq / timestamp (rtp.hip)

`step()` works on datagrams already resident in HBM.  `StreamingPipe` adds the host side of a server: pinned host buffers,
a copy-in and a copy-out HIP stream, three slots in flight, so the H2D of step k+1 and the D2H of step k-1 overlap the
compute of step k.  All arithmetic is in the HIP kernels; this file only sequences launches.
"""
import torch

from . import rtp
from .chain import ChainBatch

DATAGRAM = 172  # 12-byte RTP header + 160 G.711 codes (20 ms at 8 kHz), src/rtp.h:33, src/rtp.c:86-95
FREQ, PKT = 8000, 80


class RtpChain:
    def __init__(self, n_streams, dev, agc_value=5):
        self.n, self.dev = n_streams, dev
        # the four stages behind one C call per tick (wmx_chain_process: the AEC's far kernel beside the noise suppressor, the VAD one
        # call over the tick's two packets, as in the heartbeat)
        self.chain = ChainBatch(n_streams, 1, FREQ, 10, agc_value)
        self.snd = rtp.RtpSenders(n_streams, "a")
        self.pcm = torch.zeros((n_streams, 2 * PKT), dtype=torch.int16, device=dev)
        self.nbytes = torch.zeros(n_streams, dtype=torch.int32, device=dev)
        self.seq = torch.zeros(n_streams, dtype=torch.int16, device=dev)

    def step(self, packets_in, far, packets_out):
        """packets_in / packets_out: uint8 CUDA [n_streams, 172]; far: int16 CUDA [2, 80], the shared far-end of these 20 ms."""
        from ._lib import check, lib
        st = torch.cuda.current_stream().cuda_stream
        check(lib().wmx_rtp_ingest(self.n, packets_in.data_ptr(), packets_in.stride(0), self.pcm.data_ptr(), self.pcm.stride(0),
                                   self.nbytes.data_ptr(), self.seq.data_ptr(), st), "wmx_rtp_ingest")
        rc, _, _ = self.chain.process(far, self.pcm.view(self.n, 2, PKT))
        assert rc == 0
        return self.snd.egress(self.pcm, 1, FREQ, 1, FREQ, packets=packets_out)

    def close(self):
        for b in (self.chain, self.snd):
            b.close()


class StreamingPipe:
    """Host-resident datagrams in, host-resident datagrams out, copies overlapped with compute (3 slots in flight)."""
    SLOTS = 3

    def __init__(self, chain):
        self.c = chain
        n, dev = chain.n, chain.dev
        self.h_in = [torch.empty((n, DATAGRAM), dtype=torch.uint8).pin_memory() for _ in range(self.SLOTS)]
        self.h_out = [torch.empty((n, DATAGRAM), dtype=torch.uint8).pin_memory() for _ in range(self.SLOTS)]
        self.d_in = [torch.empty((n, DATAGRAM), dtype=torch.uint8, device=dev) for _ in range(self.SLOTS)]
        self.d_out = [torch.empty((n, DATAGRAM), dtype=torch.uint8, device=dev) for _ in range(self.SLOTS)]
        self.s_in, self.s_out = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        self.ev_in = [torch.cuda.Event() for _ in range(self.SLOTS)]
        self.ev_done = [torch.cuda.Event() for _ in range(self.SLOTS)]
        self.ev_out = [None] * self.SLOTS
        self.k = 0

    def submit(self, far):
        """Process the datagrams the caller has placed in h_in[slot] (slot = k % SLOTS); the result lands in h_out[slot]
        once ev_out[slot] has fired.  Returns the slot."""
        s = self.k % self.SLOTS
        self.k += 1
        main = torch.cuda.current_stream()
        if self.ev_out[s] is not None:
            self.ev_out[s].synchronize()  # the slot's previous result has left the device: its buffers are free
        with torch.cuda.stream(self.s_in):
            self.d_in[s].copy_(self.h_in[s], non_blocking=True)
            self.ev_in[s].record(self.s_in)
        main.wait_event(self.ev_in[s])
        self.c.step(self.d_in[s], far, self.d_out[s])
        self.ev_done[s].record(main)
        with torch.cuda.stream(self.s_out):
            self.s_out.wait_event(self.ev_done[s])
            self.h_out[s].copy_(self.d_out[s], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.s_out)
            self.ev_out[s] = ev
        return s

    def drain(self):
        for ev in self.ev_out:
            if ev is not None:
                ev.synchronize()
