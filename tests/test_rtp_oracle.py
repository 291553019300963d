"""RTP / G.711 packet edge: the oracle restatement (oracle/orc_rtp.c) against golden datagrams captured from the
real reference functions over UDP loopback (tests/golden/make_rtp_golden.py)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
from make_rtp_golden import N_PACKETS, SEND_CASES, ring_pcm  # noqa: E402

G = np.load(os.path.join(GOLDEN, "rtp_golden.npz"))


class Sender(C.Structure):
    _fields_ = [("seq", C.c_uint16), ("timestamp", C.c_uint32), ("ssrc", C.c_uint32), ("pt", C.c_uint8)]


def orc_send(port, law, chn, freq, pcm, n_packets):
    s = Sender()
    port.orc_rtp_sender_init(C.byref(s), law)
    chunk = pcm.size // n_packets
    out = []
    for k in range(n_packets):
        pkt = np.zeros(12 + 4 * chunk + 64, np.uint8)
        src = np.ascontiguousarray(pcm[k * chunk:(k + 1) * chunk])
        n = port.orc_rtp_egress(C.byref(s), 1, 8000, src.ctypes.data_as(C.c_void_p), chunk * 2, chn, freq, pkt.ctypes.data_as(C.c_void_p))
        out.append(pkt[:n].copy())
    return np.stack(out), (s.seq, s.timestamp)


@pytest.mark.parametrize("c", range(len(SEND_CASES)))
def test_egress_matches_reference_datagrams(oracle_port, c):
    chn, freq = SEND_CASES[c]
    got, (seq, ts) = orc_send(oracle_port, 0, chn, freq, ring_pcm(c, N_PACKETS), N_PACKETS)
    assert np.array_equal(got, G["send_%d" % c])
    assert seq == N_PACKETS and ts == N_PACKETS * (got.shape[1] - 12) // chn


def test_ingest_matches_reference(oracle_port):
    for k, pkt in enumerate(G["recv_in"]):
        pcm = np.zeros(160, np.int16)
        seq = C.c_uint16(0)
        p = np.ascontiguousarray(pkt)
        n = oracle_port.orc_rtp_ingest(p.ctypes.data_as(C.c_void_p), pcm.ctypes.data_as(C.c_void_p), C.byref(seq))
        assert n == G["recv_bytes"][k] and seq.value == G["recv_seq"][k]
        assert np.array_equal(pcm, G["recv_pcm"][k])


def test_sequence_wraps_and_mulaw(oracle_port):
    """uint16 sequence wrap (rtpHeader.seq++) and the mu-law variant of the same loop (pt 0, PCM2G711u)."""
    s = Sender()
    oracle_port.orc_rtp_sender_init(C.byref(s), 1)
    s.seq = 0xFFFF
    pcm = ring_pcm(1, 2)
    pkt = np.zeros(400, np.uint8)
    n = oracle_port.orc_rtp_egress(C.byref(s), 1, 8000, pcm.ctypes.data_as(C.c_void_p), 320, 1, 8000, pkt.ctypes.data_as(C.c_void_p))
    assert n == 172 and pkt[1] == 0x80 and (pkt[2], pkt[3]) == (0xFF, 0xFF) and s.seq == 0
    want = np.zeros(160, np.uint8)
    oracle_port.orc_PCM2G711u(pcm.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), 320, 0)
    assert np.array_equal(pkt[12:172], want)
