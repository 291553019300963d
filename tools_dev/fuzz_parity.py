"""fuzz_parity.py [--cases N] [--seed S] -- random shapes of the record heartbeat through wmx_chain_process against per-handle oracle
runs: channels, rate, the daemon's interval, which stages are switched on (float or fixed-point NS / AEC), batch size, packets per
call, stream-major rows with padding or packet-major steps, in place or out of place, AGC gain, reported delay.  Every case is a
shape `*_init` accepts (the others are covered by tests/test_edges_gpu.py); a case fails on the first differing sample.

The call size is part of the case: vad_process analyses packet 0 of every call, so the oracle is driven with the
same frames per call as the device.  Prints one JSON line per case and a summary; exit code 1 on any mismatch."""
import argparse
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oracle import loader as L  # noqa: E402
from wmix_amd import synth  # noqa: E402
from wmix_amd.chain import AEC, AECM, AGC, NS, NSX, VAD, ChainBatch  # noqa: E402


LAYOUTS = ("stream rows, padded per call", "tick-major: [call][stream][packets]", "packet-major: [packet][stream]", "stream rows, padded per packet")


def draw(rng):
    stages = int(rng.integers(1, 16))
    fx_ns = bool(stages & NS) and rng.random() < 0.3
    fx_aec = bool(stages & AEC) and rng.random() < 0.3
    chn = int(rng.choice([1, 1, 2]))
    freq = int(rng.choice([8000, 16000] if stages & AEC else [8000, 16000, 32000]))
    interval = int(rng.choice([10, 20]))
    per_call = int(rng.choice([1, 2, 3, 4, 6]))
    if interval == 20:
        per_call = max(2, per_call + (per_call & 1))  # whole 20 ms packets
    # a stage whose packet is longer than 10 ms (the AEC at 8 kHz and the VAD up to 16 kHz with 20 ms handles) and the VAD of
    # interleaved channels want a call's packets in one piece per stream: the API refuses the other layouts (tests/test_edges_gpu.py)
    in_one_piece = interval == 20 or (chn == 2 and bool(stages & VAD))
    layout = int(rng.integers(0, 2 if in_one_piece else 4))
    return {"stages": stages | (NSX if fx_ns else 0) | (AECM if fx_aec else 0), "chn": chn, "freq": freq, "interval_ms": interval,
            "packets_per_call": per_call, "streams": int(rng.integers(1, 97)), "calls": int(rng.integers(20, 90)),
            "layout": LAYOUTS[layout], "pad": int(rng.choice([0, 0, 8, 24])),
            "in_place": bool(rng.random() < 0.6) or stages == VAD,  # (a VAD-only chain works in place, like vad_process)
            "agc_value": int(rng.choice([5, 5, 9, 0, 30])), "delay_ms": int(rng.choice([0, 0, 0, 20, 100]))}


def run_case(c, dev, port, seed):
    chn, freq, S, P, calls, pad = c["chn"], c["freq"], c["streams"], c["packets_per_call"], c["calls"], c["pad"]
    pkt1 = freq // 100  # frames of a 10 ms packet
    pkt = pkt1 * chn
    n10 = calls * P
    far1 = synth.far_end(seed, n10, pkt1)
    near1 = synth.near_end(seed + 1, S, n10, pkt1, far=far1).reshape(S, -1)
    far, near = np.repeat(far1, chn), np.repeat(near1, chn, axis=1)
    if chn == 2:
        near[:, 1::2] = (near[:, 1::2].astype(np.int32) * 3 // 4).astype(np.int16)
    cb = ChainBatch(S, chn, freq, c["interval_ms"], c["agc_value"], c["stages"])
    dfar = torch.from_numpy(far.reshape(n10, pkt).copy()).to(dev)
    x = torch.from_numpy(near.reshape(S, calls, P, pkt)).to(dev)  # [stream][call][packet][sample]
    lay = LAYOUTS.index(c["layout"])
    # every layout as: a flat buffer, and for call k the element offset / stream stride / packet stride of its [S, P, pkt] view
    if lay == 0:
        row = P * pkt + pad
        shape, perm, geo = (S, calls, row), None, lambda k: (k * row, calls * row, pkt)
        fill = lambda buf: buf.view(S, calls, row)[:, :, :P * pkt].copy_(x.reshape(S, calls, P * pkt))  # noqa: E731
    elif lay == 1:
        row = P * pkt + pad
        shape, geo = (calls, S, row), lambda k: (k * S * row, row, pkt)
        fill = lambda buf: buf.view(calls, S, row)[:, :, :P * pkt].copy_(x.permute(1, 0, 2, 3).reshape(calls, S, P * pkt))  # noqa: E731
    elif lay == 2:
        row = pkt + pad
        shape, geo = (n10, S, row), lambda k: (k * P * S * row, row, S * row)
        fill = lambda buf: buf.view(n10, S, row)[:, :, :pkt].copy_(x.permute(1, 2, 0, 3).reshape(n10, S, pkt))  # noqa: E731
    else:
        row = pkt + pad
        shape, geo = (S, n10, row), lambda k: (k * P * row, n10 * row, row)
        fill = lambda buf: buf.view(S, n10, row)[:, :, :pkt].copy_(x.reshape(S, n10, pkt))  # noqa: E731
    n_el = int(np.prod(shape))
    buf = torch.full((n_el,), 0, dtype=torch.int16, device=dev)
    fill(buf)
    outbuf = buf if c["in_place"] else torch.full((n_el,), 0, dtype=torch.int16, device=dev)
    delays = None if c["delay_ms"] == 0 else [c["delay_ms"]]
    for k in range(calls):
        off, ss, ps = geo(k)
        pin = torch.as_strided(buf, (S, P, pkt), (ss, ps, 1), off)
        pout = None if c["in_place"] else torch.as_strided(outbuf, (S, P, pkt), (ss, ps, 1), off)
        rc, codes, _ = cb._process(dfar[k * P:(k + 1) * P], pin, pout, P, ss, ps, delays, None)
        assert rc == 0, (rc, codes)
    got = torch.stack([torch.as_strided(outbuf, (S, P, pkt), geo(k)[1:] + (1,), geo(k)[0]) for k in range(calls)], 1).cpu().numpy().reshape(S, -1)
    cb.close()
    # nothing written between the rows: the output buffer minus the packets is what it was (zero)
    check = outbuf.clone()
    for k in range(calls):
        off, ss, ps = geo(k)
        torch.as_strided(check, (S, P, pkt), (ss, ps, 1), off).zero_()
    pad_ok = not bool(check.any())
    rng = np.random.default_rng(seed)
    bad, moved = 0, 0
    for s in sorted(set(int(v) for v in rng.integers(0, S, 3))):
        want = oracle_chain(port, c, far, near[s], pkt1 * P)
        bad += int((got[s] != want).sum())
        moved += int((want != near[s]).sum())
    c["samples_the_chain_changed"] = moved  # (not a vacuous comparison: the stages did something to these streams)
    return bad, pad_ok


def oracle_chain(port, c, far, near, frames_per_call):
    """NS / NSX -> AEC / AECM -> AGC -> VAD of one handle set, call by call like the daemon (reported delay on the AEC call)."""
    st, chn, freq, iv = c["stages"], c["chn"], c["freq"], c["interval_ms"]
    x = near
    if st & NS:
        x = (L.run_nsx if st & NSX else L.run_ns)(port, chn, freq, x, frames_per_call, prefix="orc")
    if st & AEC:
        if st & AECM:
            x = L.run_aecm(port, chn, freq, iv, far, x, frames_per_call, c["delay_ms"], prefix="orc")
        else:
            x = L.run_aec(port, chn, freq, iv, far, x, frames_per_call, c["delay_ms"], prefix="orc")
    if st & AGC:
        x = L.run_agc(port, chn, freq, c["agc_value"], x, frames_per_call, prefix="orc")
    if st & VAD:
        x = L.run_vad(port, chn, freq, iv, x, frames_per_call, prefix="orc")
    return x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    dev = torch.device("cuda:0")
    port = L.port()
    failed, t0 = 0, time.time()
    for i in range(a.cases):
        c = draw(rng)
        try:
            bad, pad_ok = run_case(c, dev, port, 70000 + 97 * i + a.seed)
            c.update(case=i, samples_differing=bad, padding_untouched=pad_ok)
        except Exception as e:  # a refusal or a crash is a finding too
            c.update(case=i, error=repr(e)[:300])
            bad, pad_ok = 1, True
        failed += int(bad != 0 or not pad_ok)
        print(json.dumps(c), flush=True)
    print(json.dumps({"cases": a.cases, "failed": failed, "seed": a.seed, "wall_s": round(time.time() - t0, 1)}))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
