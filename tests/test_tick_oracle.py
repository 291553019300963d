"""CPU: the daemon's whole tick -- task threads' wmix_load_data, the play thread's package, playPkgBuff_add / _get, the record
heartbeat NS -> AEC(far = the delayed playback) -> AGC -> VAD, the zoom to 1 x 8000 (src/wmix.c:1347-1440 with :528-780 inside) --
composed from the restatement (oracle.loader.tick_port) against the same composition of the REAL functions as compiled from
/root/reference (oracle/_ref/ref_mix_driver tick).  Every stage's data is compared, bit for bit: the float stages of the restatement
are bit-exact against the generic-C reference on the same inputs (tests/test_ns_oracle.py, test_aec_oracle.py)."""
import numpy as np
import pytest

from oracle import loader as L

pytestmark = pytest.mark.skipif(not L.have_ref_mix(), reason="oracle/_ref/ref_mix_driver not built (no /root/reference here)")


def tick_inputs(seed, T, n_src, n_rec, src_freq, src_chn, loud=8000):
    """sources int16 [T, n_src, 20 ms of (src_freq, src_chn)], local microphone signal int16 [T, n_rec, 160]"""
    rng = np.random.default_rng(seed)
    fr = src_freq // 1000 * 20
    t = np.arange(T * fr)
    src = np.zeros((T, n_src, fr * src_chn), np.int16)
    for i in range(n_src):
        tone = loud * np.sin(2 * np.pi * (200 + 61 * i) * t / src_freq) * (((t // (fr * 10)) + i) % 3 > 0)  # on / off every 200 ms
        x = np.clip(tone + rng.integers(-1500, 1500, t.size), -32768, 32767).astype(np.int16)
        cols = [x] + [x // 3] * (src_chn - 1)
        src[:, i] = np.stack(cols, 1).reshape(T, fr * src_chn)
    tt = np.arange(T * 160)
    local = np.zeros((T, n_rec, 160), np.int16)
    for k in range(n_rec):
        speech = 3000 * np.sin(0.01 * (1 + 0.1 * k) * tt) * ((tt // 16000 + k) % 2)  # a talker, one second on, one off
        local[:, k] = (speech + rng.integers(-200, 200, tt.size)).astype(np.int16).reshape(T, 160)
    return src, local


@pytest.mark.parametrize("src_freq,src_chn,n_src,n_rec,stages,T", [
    (32000, 2, 4, 2, 15, 160),   # configs[4]'s sources into the shipped ring, the whole heartbeat
    (8000, 1, 3, 1, 15, 120),    # sources already in the ring's format
    (16000, 1, 2, 1, 15, 100),
    (44100, 2, 2, 1, 2, 80),     # a rate that does not divide: AEC alone
    (8000, 1, 8, 1, 1 | 2, 100),  # eight loud sources: the saturating accumulate is order dependent
])
def test_port_tick_equals_the_real_functions(oracle_port, src_freq, src_chn, n_src, n_rec, stages, T):
    src, local = tick_inputs(7 + n_src, T, n_src, n_rec, src_freq, src_chn, loud=14000 if n_src == 8 else 8000)
    a = L.tick_port(oracle_port, src, local, src_freq, src_chn, stages=stages)
    b = L.tick_ref(src, local, src_freq, src_chn, stages=stages)
    for k in ("play", "far", "out", "zoom"):
        assert np.array_equal(a[k], b[k]), k
    # the composition is what it says: every far-end package is a package that was played earlier (WHICH one is the reference's
    # own index arithmetic, src/wmix.c:494-510: the oldest slot of the 22 while the cursor is below 20, slot 20 itself for the two
    # ticks it is not -- tests/test_pkgfifo.py pins that against the real functions) or the FIFO's initial silence ...
    for t in range(T):
        assert any(np.array_equal(a["far"][t], a["play"][u]) for u in range(max(0, t - 22), t + 1)) or not a["far"][t].any(), t
    # ... the playback starts VIEW_PLAY_CORRECT = 200 ms after the first load, and the canceller had something to cancel
    assert not a["play"][:10].any() and a["play"][10:].any()
    if stages & 2:
        assert np.abs(a["out"][60:].astype(np.int32)).mean() < np.abs(a["near"][60:].astype(np.int32)).mean()
