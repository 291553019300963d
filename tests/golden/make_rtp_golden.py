"""Generates tests/golden/rtp_golden.npz from the REAL reference: oracle/_ref/ref_mix_driver runs the loop body of
wmix_thread_rtp_send_pcma (wmix_pcm_zoom -> PCM2G711a -> rtp_send) and rtp_recv -> G711a2PCM over UDP loopback.
Run in the build container:  python tests/golden/make_rtp_golden.py"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import loader  # noqa: E402

# target formats of the sender (the daemon's ring is 1 x 8000, platform/alsa/plat.h:48-50); 20 ms per packet
SEND_CASES = [(1, 8000), (2, 8000), (1, 16000)]
N_PACKETS = 40


def ring_pcm(case, n_packets):
    """20 ms chunks of the 1 x 8000 ring: tone + LCG noise"""
    n = 160 * n_packets
    x = np.uint32(9000 + case)
    v = np.zeros(n, np.int64)
    for i in range(n):
        x = np.uint32((int(x) * 1664525 + 1013904223) & 0xFFFFFFFF)
        v[i] = ((int(x) >> 16) % 6001) - 3000
    t = np.arange(n)
    return np.clip(v + 9000 * np.sin(2 * np.pi * 440 * t / 8000), -32768, 32767).astype(np.int16)


def split_records(blob):
    out, off = [], 0
    while off < len(blob):
        (n,) = struct.unpack_from("<I", blob, off)
        out.append(blob[off + 4: off + 4 + n])
        off += 4 + n
    return out


def main():
    g = {}
    for c, (chn, freq) in enumerate(SEND_CASES):
        pcm = ring_pcm(c, N_PACKETS)
        wire = split_records(loader.ref_mix("rtpsend", chn, freq, stdin=pcm.tobytes()))
        assert len(wire) == N_PACKETS and len({len(w) for w in wire}) == 1
        g["send_%d" % c] = np.frombuffer(b"".join(wire), np.uint8).reshape(N_PACKETS, -1)
    # receive side: the 1 x 8000 packets above, plus a mu-law-tagged, an unknown-type and a high-sequence packet
    pk = [bytes(p) for p in g["send_0"]]
    odd = [bytearray(pk[1]), bytearray(pk[2]), bytearray(pk[3])]
    odd[0][1] = 0x80 | 0   # PCMU tag: rtp_recv still says 160 bytes, the thread still decodes A-law
    odd[1][1] = 0x80 | 9   # G722: size 0
    odd[2][2], odd[2][3] = 0xAB, 0xCD
    recv_in = pk[:8] + [bytes(o) for o in odd]
    blob = loader.ref_mix("rtprecv", stdin=b"".join(struct.pack("<I", len(p)) + p for p in recv_in))
    pcm_out, sizes, seqs, off = [], [], [], 0
    for _ in recv_in:
        (n,) = struct.unpack_from("<I", blob, off)
        off += 4
        row = np.zeros(160, np.int16)
        row[: n // 2] = np.frombuffer(blob[off: off + n], np.int16)
        off += n
        (s,) = struct.unpack_from("<H", blob, off)
        off += 2
        pcm_out.append(row)
        sizes.append(n)
        seqs.append(s)
    g["recv_in"] = np.frombuffer(b"".join(recv_in), np.uint8).reshape(len(recv_in), -1)
    g["recv_pcm"] = np.stack(pcm_out)
    g["recv_bytes"] = np.array(sizes, np.uint32)
    g["recv_seq"] = np.array(seqs, np.uint16)
    np.savez_compressed(os.path.join(HERE, "rtp_golden.npz"), **g)
    print("wrote", {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
