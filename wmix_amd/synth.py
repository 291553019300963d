"""Synthetic 10 ms frame generators shared by bench.py and the tests.

Recipe from BASELINE.md section 3 / SURVEY.md section 8d: a 32-bit LCG
(x = x*1664525 + 1013904223) per stream seeded seed0 + stream_id, sample =
((x >> 16) % (2A+1)) - A; near = far delayed 40 samples / 2 + noise(A=200) + a
3000*sin(0.01 t) tone gated on/off every 100 frames; shared far = noise A=8000.
Pure numpy (vectorised jump-ahead LCG), no reference code involved.
"""
import numpy as np

_A = np.uint32(1664525)
_C = np.uint32(1013904223)


def lcg_states(seeds, n):
    """Return uint32 array [len(seeds), n]: n successive LCG states after each seed."""
    seeds = np.atleast_1d(np.asarray(seeds, dtype=np.uint64)).astype(np.uint32)
    with np.errstate(over="ignore"):
        # x_k = A^k x_0 + C * (1 + A + ... + A^(k-1))   (mod 2^32, uint32 wraps)
        apow = np.empty(n + 1, dtype=np.uint32)
        apow[0] = 1
        apow[1:] = _A
        apow = np.multiply.accumulate(apow, dtype=np.uint32)
        geo = np.add.accumulate(apow[:-1], dtype=np.uint32)  # sum_{j<k} A^j, k=1..n
        return (apow[1:][None, :] * seeds[:, None] + (_C * geo)[None, :]).astype(np.uint32)


def lcg_noise(seeds, n, amp):
    """int16 noise [len(seeds), n] uniform in [-amp, amp]."""
    x = lcg_states(seeds, n)
    return (((x >> np.uint32(16)) % np.uint32(2 * amp + 1)).astype(np.int32) - amp).astype(np.int16)


def gated_tone(n_frames, pkt, amp=3000.0, w=0.01, period=100):
    t = np.arange(n_frames * pkt, dtype=np.float64)
    gate = ((np.arange(n_frames) // period) % 2 == 0).repeat(pkt)
    return (amp * np.sin(w * t) * gate)


def far_end(seed, n_frames, pkt, amp=8000):
    """Shared far-end reference, mono int16 [n_frames*pkt]."""
    return lcg_noise([seed], n_frames * pkt, amp)[0]


def near_end(seed0, n_streams, n_frames, pkt, far=None, delay=40, noise_amp=200, tone=True):
    """Near-end microphone signals int16 [n_streams, n_frames*pkt]."""
    n = n_frames * pkt
    seeds = seed0 + np.arange(n_streams, dtype=np.uint64)
    x = lcg_noise(seeds, n, noise_amp).astype(np.float64)
    if far is not None:
        d = np.zeros(n, dtype=np.float64)
        d[delay:] = far[: n - delay]
        x += np.trunc(d / 2.0)
    if tone:
        x += np.trunc(gated_tone(n_frames, pkt))[None, :]
    return np.clip(x, -32768, 32767).astype(np.int16)


def ns_input(seed0, n_streams, n_frames, pkt, noise_amp=3000):
    """NS-only config: noise A=3000 + gated tone. int16 [n_streams, n_frames*pkt]."""
    return near_end(seed0, n_streams, n_frames, pkt, far=None, noise_amp=noise_amp, tone=True)


def conference_inputs(seed, n_ticks, n_src, n_rec, src_freq, src_chn, loud=8000, tick_ms=20):
    """One mixer's inputs for the daemon's tick (wmix_amd.tick / oracle.loader.tick_port): sources int16 [n_ticks, n_src, tick_ms
    of (src_freq, src_chn) interleaved] -- a tone per source, on and off every 200 ms, in noise, the right channel a third of the
    left -- and the microphones' LOCAL signal int16 [n_ticks, n_rec, 8000 / 1000 * tick_ms]: a talker one second on, one off, in
    noise A = 200 (what the microphone picks up without the loudspeaker; the room adds the echo)."""
    rng = np.random.default_rng(seed)
    fr = src_freq // 1000 * tick_ms
    t = np.arange(n_ticks * fr)
    src = np.zeros((n_ticks, n_src, fr * src_chn), np.int16)
    for i in range(n_src):
        tone = loud * np.sin(2 * np.pi * (200 + 61 * i) * t / src_freq) * (((t // (fr * 10)) + i) % 3 > 0)
        x = np.clip(tone + rng.integers(-1500, 1500, t.size), -32768, 32767).astype(np.int16)
        cols = [x] + [x // 3] * (src_chn - 1)
        src[:, i] = np.stack(cols, 1).reshape(n_ticks, fr * src_chn)
    n = 8 * tick_ms
    tt = np.arange(n_ticks * n)
    local = np.zeros((n_ticks, n_rec, n), np.int16)
    for k in range(n_rec):
        speech = 3000 * np.sin(0.01 * (1 + 0.1 * k) * tt) * ((tt // 16000 + k) % 2)
        local[:, k] = (speech + rng.integers(-200, 200, tt.size)).astype(np.int16).reshape(n_ticks, n)
    return src, local
