"""Generates tests/golden/g711_golden.npz from the REAL reference
(oracle/_ref/libwmixref.so = /root/reference/src/g711codec.c compiled as is).

Contents: the complete known-answer tables of both companders (every int16 ->
code, every code -> int16), the return values of the four PCM2G711x/G711x2PCM
calls, and FNV-1a-64 hashes + a 1 s excerpt of the codec run over the
reference's own test asset audio/1x8000.wav in 80-sample frames (config #1).
Run in the build container only:  python tests/golden/make_g711_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import loader as L  # noqa: E402

ref = L.ref()
allpcm = np.arange(-32768, 32768, dtype=np.int16)
codes = np.arange(256, dtype=np.uint8)
out = {}
for law in "au":
    e, r = L.g711_encode(ref, law, allpcm)
    assert r == 65536
    d, r = L.g711_decode(ref, law, codes)
    assert r == 512
    out["enc_" + law] = e
    out["dec_" + law] = d
raw = open("/root/reference/audio/1x8000.wav", "rb").read()
pcm = np.frombuffer(raw[44:44 + 6078 * 80 * 2], dtype=np.int16)
out["wav_excerpt"] = pcm[8000:16000].copy()  # 1 s of real speech, 100 frames of 80
for law in "au":
    e, _ = L.g711_encode(ref, law, pcm)
    d, _ = L.g711_decode(ref, law, e)
    out["wav_hash_enc_" + law] = np.uint64(L.fnv1a64(e.tobytes()))
    out["wav_hash_dec_" + law] = np.uint64(L.fnv1a64(d.tobytes()))
    out["wav_excerpt_enc_" + law] = e[8000:16000].copy()
    out["wav_excerpt_dec_" + law] = d[8000:16000].copy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g711_golden.npz"), **out)
print({k: (v.shape if hasattr(v, "shape") and v.shape else v) for k, v in out.items()})
