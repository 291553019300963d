"""developer check: where a 2-LSB sample of the float chain's output comes from.  The float AEC may differ from the reference by 1 LSB on
isolated samples (powf / cosf / sinf of the host's libm are not always correctly rounded; tests/test_aec_gpu.py check_float_path); the
AGC behind it multiplies by its gain (compression 5 dB: up to x 1.78), so a 1-LSB difference going in can be 2 LSB coming out.  This
runs S lockstep streams for T packets and reports
  (a) GPU NS -> AEC            against the oracle's NS -> AEC          (the tolerance stage: <= 1 LSB expected)
  (b) GPU NS -> AEC -> AGC -> VAD against the oracle's whole chain       (what a long soak sees: rare 2-LSB samples)
  (c) the ORACLE's AGC -> VAD applied to the GPU's NS -> AEC output, against the GPU's own AGC -> VAD output   (must be 0: the integer
      stages are exact on the input they are given)

    python tools_dev/chain_lsb_attribution.py [--streams 256] [--packets 6000] [--freq 16000]
"""
import argparse
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import loader as L  # noqa: E402
from wmix_amd import synth  # noqa: E402
from wmix_amd.chain import AEC, AGC, NS, VAD, ChainBatch  # noqa: E402


def run(dev, S, T, freq, stages, far, inp):
    pkt = freq // 100
    cb = ChainBatch(S, 1, freq, 10, 5, stages=stages, n_cohorts=1)
    out = torch.empty_like(inp)
    for t in range(T):
        rc, codes, _ = cb.process_packet_major(far[t:t + 1], inp[t:t + 1], out=out[t:t + 1])
        assert rc == 0
    cb.close()
    return out.cpu().numpy().transpose(1, 0, 2).reshape(S, T * pkt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--packets", type=int, default=6000)
    ap.add_argument("--freq", type=int, default=16000)
    a = ap.parse_args()
    S, T, freq = a.streams, a.packets, a.freq
    pkt = freq // 100
    dev = torch.device("cuda:0")
    far = synth.far_end(9100, T, pkt).reshape(T, pkt)
    near = synth.near_end(9101, S, T, pkt, far=far.reshape(-1)).reshape(S, T, pkt)
    dfar = torch.from_numpy(far.copy()).to(dev)
    inp = torch.from_numpy(np.ascontiguousarray(near.transpose(1, 0, 2))).to(dev)
    g_aec = run(dev, S, T, freq, NS | AEC, dfar, inp)
    g_all = run(dev, S, T, freq, NS | AEC | AGC | VAD, dfar, inp)
    port = L.port()
    h = {"aec": np.zeros(4, np.int64), "chain": np.zeros(4, np.int64), "tail_on_gpu_aec": np.zeros(4, np.int64)}
    n = 0
    for s in range(S):
        o_ns = L.run_ns(port, 1, freq, near[s].reshape(-1), pkt, prefix="orc")
        o_aec = L.run_aec(port, 1, freq, 10, far.reshape(-1), o_ns, pkt, 0, prefix="orc")
        o_all = L.run_vad(port, 1, freq, 10, L.run_agc(port, 1, freq, 5, o_aec, pkt, prefix="orc"), pkt, prefix="orc")
        o_tail = L.run_vad(port, 1, freq, 10, L.run_agc(port, 1, freq, 5, g_aec[s], pkt, prefix="orc"), pkt, prefix="orc")
        for key, got, want in (("aec", g_aec[s], o_aec), ("chain", g_all[s], o_all), ("tail_on_gpu_aec", g_all[s], o_tail)):
            d = np.abs(got.astype(np.int32) - want.astype(np.int32))
            h[key] += np.bincount(np.minimum(d, 3), minlength=4)
        n += o_all.size
    print(json.dumps({"streams": S, "packets": T, "freq": freq, "samples": n,
                      "abs_diff_histogram_0_1_2_3plus": {k: v.tolist() for k, v in h.items()}}))
    assert h["aec"][2:].sum() == 0 and h["tail_on_gpu_aec"][1:].sum() == 0 and h["chain"][3] == 0


if __name__ == "__main__":
    main()
