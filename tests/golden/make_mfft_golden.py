"""Generates tests/golden/mfft_golden.npz from the REAL reference math/fft.c (oracle/_ref/libwmixref.so, built by
oracle/Makefile from /root/reference/math/fft.c).  Run in the build container:  python tests/golden/make_mfft_golden.py

The reference has no tests or vectors for math/fft.c (and no callers), so these are outputs of the
reference itself on seeded inputs: every transform kind at N = 2 .. 2048, a 1024-point fft_stream run, and the
NULL-argument forms."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import loader  # noqa: E402

SIZES = [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048]


def mfft_input(n, seed):
    """tone + LCG noise, amplitudes of int16 audio (what fft_stream is fed in the reference's demo use)"""
    x = np.uint32(seed)
    out = np.zeros((2, n), np.float32)
    t = np.arange(n)
    for c in range(2):
        v = np.zeros(n, np.int64)
        for i in range(n):
            x = np.uint32((int(x) * 1664525 + 1013904223) & 0xFFFFFFFF)
            v[i] = ((int(x) >> 16) % 4001) - 2000
        out[c] = (v + np.round(6000 * np.sin(2 * np.pi * (3 + c) * t / max(n, 8)))).astype(np.float32)
    return out[0], out[1]


def main():
    ref = loader.ref()
    g = {}
    for n in SIZES:
        re, im = mfft_input(n, 7000 + n)
        for kind in range(4):
            o = loader.mfft(ref, kind, re, im, n)
            for k, v in o.items():
                g["k%d_n%d_%s" % (kind, n, k)] = v
    # NULL forms at N = 256: no imaginary input; no input at all
    re, _ = mfft_input(256, 7777)
    for kind in range(4):
        o = loader.mfft(ref, kind, re, None, 256)
        for k, v in o.items():
            g["k%d_noim_%s" % (kind, k)] = v
    # fft_stream: 1024-sample pool fed 160 samples at a time, 12 calls
    sig, _ = mfft_input(160 * 12, 7100)
    stream, afs, pfs = loader.mfft_stream(ref, sig.reshape(12, 160), 1024)
    g["stream_final"] = stream
    g["stream_af"] = np.stack(afs)
    g["stream_pf"] = np.stack(pfs)
    np.savez_compressed(os.path.join(HERE, "mfft_golden.npz"), **g)
    print("wrote", len(g), "arrays")


if __name__ == "__main__":
    main()
