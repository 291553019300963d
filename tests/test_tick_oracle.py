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
    from wmix_amd import synth
    return synth.conference_inputs(seed, T, n_src, n_rec, src_freq, src_chn, loud=loud)


@pytest.mark.parametrize("src_freq,src_chn,n_src,n_rec,stages,T", [
    (32000, 2, 4, 2, 15, 160),   # configs[4]'s sources into the shipped ring, the whole heartbeat
    (8000, 1, 3, 1, 15, 120),    # sources already in the ring's format
    (16000, 1, 2, 1, 15, 100),
    (44100, 2, 2, 1, 2, 80),     # a rate that does not divide: AEC alone
    (8000, 1, 8, 1, 1 | 2, 100),  # eight loud sources: the saturating accumulate is order dependent
    (16000, 2, 3, 2, 15 | 16, 130),  # WR_NS_PA: the playback goes through ns_process in front of the FIFO (src/wmix.c:1370-1386)
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
    if stages & 16:  # the suppressor changed what was played (and so what the cancellers heard)
        plain = L.tick_port(oracle_port, src, local, src_freq, src_chn, stages=stages & 15)
        assert not np.array_equal(plain["play"], a["play"]) and not np.array_equal(plain["out"], a["out"])
    if stages & 2:
        assert np.abs(a["out"][60:].astype(np.int32)).mean() < np.abs(a["near"][60:].astype(np.int32)).mean()


@pytest.mark.parametrize("platform", ["hi3516", "t31"])
@pytest.mark.parametrize("src_freq,src_chn,n_src,n_rec,stages,T", [(32000, 2, 3, 2, 15, 150), (8000, 1, 2, 1, 15 | 16, 110)])
def test_port_tick_equals_the_other_platform_builds(oracle_port, platform, src_freq, src_chn, n_src, n_rec, stages, T):
    """The hi3516 and t31 daemons (platform/<name>/plat.h:10-16): far-end 700 ms / 0 ms behind the playback, 37 / 2 FIFO slots, and a
    fresh source lands AT the play head (PLAT_PLAY_CORRECT 0).  The whole tick of each build against the restatement."""
    if not L.have_ref_mix(platform):
        pytest.skip("oracle/_ref/ref_mix_driver_%s not present" % platform)
    aec_ms, correct = L.PLATFORMS[platform]
    src, local = tick_inputs(31 + n_src, T, n_src, n_rec, src_freq, src_chn)
    a = L.tick_port(oracle_port, src, local, src_freq, src_chn, stages=stages, aec_delay_ms=aec_ms, play_correct=correct)
    b = L.tick_ref(src, local, src_freq, src_chn, stages=stages, platform=platform)
    for k in ("play", "far", "out", "zoom"):
        assert np.array_equal(a[k], b[k]), k
    assert a["play"][0].any()  # no 200 ms of lead: the first package already plays
    if aec_ms == 0:  # slot 0 of 2, refreshed every second tick (tests/test_pkgfifo.py): this tick's package, or the previous one
        assert np.array_equal(a["far"], a["play"][np.arange(T) & ~1])
    alsa = L.tick_port(oracle_port, src, local, src_freq, src_chn, stages=stages)
    assert not np.array_equal(alsa["out"], a["out"])


@pytest.mark.parametrize("platform", ["alsa", "t31"])
def test_port_tick_with_the_self_send_receive_test(oracle_port, platform):
    """wmix->rwTest (src/wmix.c:714-732): the heartbeat loads what it recorded back into the play ring with a cursor of its own --
    a loop through loudspeaker, FIFO, room and cancellers.  Stage bit 32 of both compositions; two record handle sets, of which the
    first feeds the ring (the daemon has one)."""
    if not L.have_ref_mix(platform):
        pytest.skip("oracle/_ref/ref_mix_driver_%s not present" % platform)
    aec_ms, correct = L.PLATFORMS[platform]
    T, n_src, n_rec = 140, 2, 2
    src, local = tick_inputs(77, T, n_src, n_rec, 16000, 1)
    src[40:] = 0  # the task threads fall silent: from then on the loudspeaker plays the recording alone
    a = L.tick_port(oracle_port, src, local, 16000, 1, stages=15 | 32, aec_delay_ms=aec_ms, play_correct=correct)
    b = L.tick_ref(src, local, 16000, 1, stages=15 | 32, platform=platform)
    for k in ("play", "far", "out", "zoom"):
        assert np.array_equal(a[k], b[k]), k
    plain = L.tick_port(oracle_port, src, local, 16000, 1, stages=15, aec_delay_ms=aec_ms, play_correct=correct)
    assert not plain["play"][60:].any() and a["play"][60:].any()
    # the heartbeat runs behind the drain of its tick: what it loads is played one package + VIEW_PLAY_CORRECT later
    assert np.array_equal(a["play"][50 + 1 + correct // 320], a["out"][50, 0])
