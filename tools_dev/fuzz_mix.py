"""fuzz_mix.py [--cases N] [--seed S] -- random formats through wmx_pcm_zoom and wmx_mix_load against the restatement (which is pinned
on the real wmix_pcm_zoom / wmix_load_data for the shipped ring): channel counts, odd rate pairs, lengths, batches with padded rows;
ring formats, source formats, reduce modes, several sources per call, play heads anywhere in the ring (the wrap included), several
groups.  Prints a summary line; exit code 1 on any mismatch."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
from oracle import loader as L  # noqa: E402
from wmix_amd._lib import WmxError  # noqa: E402
from wmix_amd.mix import MixBatch, pcm_zoom  # noqa: E402

RATES = [5000, 8000, 11025, 12000, 16000, 22050, 24000, 32000, 44100, 48000]


def zoom_case(rng, dev, port):
    ic, oc = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    ifr, ofr = int(rng.choice(RATES)), int(rng.choice(RATES))
    frames = int(rng.integers(1, 700))
    S = int(rng.integers(1, 40))
    x = rng.integers(-32768, 32768, size=(S, frames * ic), dtype=np.int16)
    pad = int(rng.choice([0, 0, 6]))
    d = torch.zeros((S, frames * ic + pad), dtype=torch.int16, device=dev)
    d[:, : frames * ic] = torch.from_numpy(x).to(dev)
    got = pcm_zoom(ic, ifr, d[:, : frames * ic], oc, ofr).cpu().numpy()
    bad = 0
    for s in sorted(set(int(v) for v in rng.integers(0, S, 3))):
        want = L.mix_zoom(port, ic, ifr, x[s], oc, ofr)
        bad += int(want.size != got.shape[1]) or int((got[s] != want).sum())
    return {"kind": "zoom", "in": [ic, ifr], "out": [oc, ofr], "frames": frames, "streams": S, "bad": bad}


def load_case(rng, dev, port):
    ring_chn, ring_freq = int(rng.integers(1, 3)), int(rng.choice([8000, 8000, 16000, 22050, 44100]))
    chn, freq = int(rng.integers(1, 3)), int(rng.choice(RATES))
    rmode, rarg = int(rng.choice([1, 1, 2, 4])), int(rng.choice([1, 1, 2, 4]))
    nsrc = int(rng.integers(1, 7))
    frames = int(rng.integers(1, 500))
    sbytes = frames * chn * 2
    size = ring_chn * 2 * ring_freq
    start = int(rng.choice([0, size - 2 * ring_chn, int(rng.integers(0, size // (2 * ring_chn))) * 2 * ring_chn]))
    G = int(rng.choice([1, 1, 3]))
    correct = None if rng.random() < 0.7 else 0
    src = rng.integers(-30000, 30000, size=nsrc * sbytes // 2 + 8, dtype=np.int16)
    c = {"kind": "load", "ring": [ring_chn, ring_freq], "src": [chn, freq], "reduce": [rmode, rarg], "sources": nsrc, "bytes": sbytes,
         "start": start, "groups": G, "play_correct": correct}
    want, meta = L.mix_load(port, ring_chn, ring_freq, freq, chn, rmode, rarg, nsrc, sbytes, start, src, play_correct=correct)
    mb = MixBatch(G, ring_chn, ring_freq)
    try:
        mb.set(start, 0, rmode)
        if correct is not None:
            mb.set_play_correct(correct)
        per = sbytes // 2
        d = torch.from_numpy(np.ascontiguousarray(np.tile(src[None, :], (G, 1)))).to(dev)
        view = torch.as_strided(d, (G, nsrc, per + chn), (d.stride(0), per, 1))
        try:
            h, t = mb.load(view, sbytes, freq, chn, reduce=rarg)
        except WmxError as e:
            c["refused"] = str(e)[-90:]
            c["bad"] = 0  # (the library refuses what the reference would overrun: more than 64 fill samples, more than one ring)
            return c
        bad = 0 if (t, h) == tuple(int(v) for v in meta[-1]) else 1
        for g in range(G):
            bad += int((mb.export(g)[0] != want).sum())
        c["bad"] = bad
    finally:
        mb.close()
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    dev = torch.device("cuda:0")
    port = L.port()
    failed, refused, n = 0, 0, {"zoom": 0, "load": 0}
    for i in range(a.cases):
        c = (zoom_case if i % 2 == 0 else load_case)(rng, dev, port)
        n[c["kind"]] += 1
        refused += int("refused" in c)
        if c["bad"]:
            failed += 1
            print(json.dumps(c), flush=True)
    print(json.dumps({"cases": a.cases, "zoom": n["zoom"], "load": n["load"], "refused_by_the_library": refused, "failed": failed, "seed": a.seed}))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
