"""CPU: oracle/orc_mix.c against golden results of the real wmix_pcm_zoom / wmix_len_of_* / wmix_load_data
(tests/golden/mix_golden.npz) and against oracle/_ref/ref_mix_driver on fresh inputs when present.  Bit-exact."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import loader as L

sys.path.insert(0, GOLDEN)
from make_mix_golden import LOAD_CASES, ZOOM_CASES, load_input, zoom_input  # noqa: E402

G = np.load(os.path.join(GOLDEN, "mix_golden.npz"))


Ring, _bind, orc_zoom, orc_load = L.MixRing, L.mix_bind, L.mix_zoom, L.mix_load  # the helpers live beside the other oracle drivers


@pytest.mark.parametrize("i", range(len(ZOOM_CASES)))
def test_zoom_and_len_golden(oracle_port, i):
    _bind(oracle_port)
    ic, ifr, oc, ofr, n = ZOOM_CASES[i]
    x = zoom_input(i, n)
    assert np.array_equal(orc_zoom(oracle_port, ic, ifr, x, oc, ofr), G["zoom_%d" % i])
    assert oracle_port.orc_len_of_out(ic, ifr, n, oc, ofr) == G["lens_%d" % i][0]
    assert oracle_port.orc_len_of_in(ic, ifr, oc, ofr, n) == G["lens_%d" % i][1]


def test_survey_smoke_values(oracle_port):
    """zoom(2,32000,2560 B -> 1,8000) = 320 B; (2,32000 -> 2,16000) = 0 B (dead branch);
    load of 1280 int16 of 2x32k into a fresh 1x8000 ring -> tick 3520, head 3520."""
    _bind(oracle_port)
    x = zoom_input(0, 2560)
    assert orc_zoom(oracle_port, 2, 32000, x, 1, 8000).size * 2 == 320
    assert orc_zoom(oracle_port, 2, 32000, x, 2, 16000).size == 0
    src = load_input(0, 1, 2560)
    _, meta = orc_load(oracle_port, 1, 8000, 32000, 2, 1, 1, 1, 2560, 0, src)
    assert tuple(meta[0]) == (3520, 3520)


@pytest.mark.parametrize("i", range(len(LOAD_CASES)))
def test_load_data_golden(oracle_port, i):
    _bind(oracle_port)
    freq, chn, rmode, rarg, nsrc, sbytes, start = LOAD_CASES[i]
    ring, meta = orc_load(oracle_port, 1, 8000, freq, chn, rmode, rarg, nsrc, sbytes, start, load_input(i, nsrc, sbytes))
    assert np.array_equal(ring, G["ring_%d" % i]) and np.array_equal(meta, G["meta_%d" % i])


def test_saturation_is_order_dependent(oracle_port):
    """volumeAdd saturates per step (src/wmix.c:1617-1636): (+30000) + (+30000) + (-30000) != (-30000) + ... order."""
    _bind(oracle_port)
    a = np.full(84, 30000, np.int16)
    b = np.full(84, -30000, np.int16)
    r1, _ = orc_load(oracle_port, 1, 8000, 8000, 1, 1, 1, 3, 160, 0, np.concatenate([a[:80], a[:80], b]))
    r2, _ = orc_load(oracle_port, 1, 8000, 8000, 1, 1, 1, 3, 160, 0, np.concatenate([b[:80], a[:80], a]))
    assert r1[1600] == 32767 - 30000 and r2[1600] == 30000  # head starts 3200 B = 1600 samples ahead (VIEW_PLAY_CORRECT)


def test_against_real_reference_fresh_inputs(oracle_port):
    if not L.have_ref_mix():
        pytest.skip("oracle/_ref/ref_mix_driver not present")
    _bind(oracle_port)
    rng = np.random.default_rng(1234)
    for (ic, ifr, oc, ofr) in ((2, 32000, 1, 8000), (1, 11025, 2, 44100), (2, 16000, 1, 48000), (1, 48000, 1, 8000)):
        for n in (4, 90, 1764, 4096):
            x = rng.integers(-32768, 32768, size=n // 2, dtype=np.int16)
            want = np.frombuffer(L.ref_mix("zoom", ic, ifr, oc, ofr, stdin=x.tobytes()), dtype=np.int16)
            assert np.array_equal(orc_zoom(oracle_port, ic, ifr, x, oc, ofr), want)
    for (freq, chn) in ((8000, 1), (32000, 2), (44100, 1), (5000, 2), (12000, 1)):
        for (rmode, rarg, nsrc, sbytes, start) in ((1, 1, 5, 1000, 0), (4, 1, 3, 2000, 15000), (2, 2, 2, 400, 8000)):
            src = rng.integers(-25000, 25000, size=nsrc * sbytes // 2 + 8, dtype=np.int16)
            b = L.ref_mix("load", freq, chn, rmode, rarg, nsrc, sbytes, start, stdin=src.tobytes())
            ring, meta = orc_load(oracle_port, 1, 8000, freq, chn, rmode, rarg, nsrc, sbytes, start, src)
            assert np.array_equal(ring, np.frombuffer(b[:16000], dtype=np.int16))
            assert np.array_equal(meta, np.frombuffer(b[16000:], dtype=np.uint32).reshape(nsrc, 2))


@pytest.mark.parametrize("platform", ["hi3516", "t31"])
def test_against_the_other_platform_builds(oracle_port, platform):
    """The reference's hi3516 and t31 builds (platform/<name>/plat.h:10-16) place a fresh source AT the play head:
    PLAT_PLAY_CORRECT is 0 there, 3200 bytes in platform/alsa (src/wmix.c:1668-1669).  src/wmix.c compiled against each header
    (oracle/Makefile) vs the restatement with that constant."""
    if not L.have_ref_mix(platform):
        pytest.skip("oracle/_ref/ref_mix_driver_%s not present" % platform)
    _bind(oracle_port)
    aec_ms, correct = L.PLATFORMS[platform]
    assert L.ref_mix("consts", platform=platform).split() == [b"1", b"8000", b"16000", str(correct).encode()]
    rng = np.random.default_rng(4321)
    for (freq, chn) in ((8000, 1), (32000, 2), (44100, 1), (12000, 1)):
        for (rmode, rarg, nsrc, sbytes, start) in ((1, 1, 5, 1000, 0), (4, 1, 3, 2000, 15000), (2, 2, 2, 400, 15998)):
            src = rng.integers(-25000, 25000, size=nsrc * sbytes // 2 + 8, dtype=np.int16)
            b = L.ref_mix("load", freq, chn, rmode, rarg, nsrc, sbytes, start, stdin=src.tobytes(), platform=platform)
            ring, meta = L.mix_load(oracle_port, 1, 8000, freq, chn, rmode, rarg, nsrc, sbytes, start, src, play_correct=correct)
            assert np.array_equal(ring, np.frombuffer(b[:16000], dtype=np.int16))
            assert np.array_equal(meta, np.frombuffer(b[16000:], dtype=np.uint32).reshape(nsrc, 2))
            other, _ = L.mix_load(oracle_port, 1, 8000, freq, chn, rmode, rarg, nsrc, sbytes, start, src)  # platform/alsa's 3200
            assert not np.array_equal(other, ring)


def test_random_formats_against_the_real_reference(oracle_port):
    """Seeded random campaign, restatement vs the REAL wmix_pcm_zoom / wmix_load_data (ref_mix_driver): odd rate pairs, both channel
    counts, lengths, reduce modes, several sources, play heads anywhere in the ring (the wrap included).  tools_dev/fuzz_mix.py draws
    the same kind of cases for the device against the restatement."""
    if not L.have_ref_mix():
        pytest.skip("oracle/_ref/ref_mix_driver not present")
    _bind(oracle_port)
    rng = np.random.default_rng(int(os.environ.get("WMIX_FUZZ_SEED", "77")))
    rates = [5000, 8000, 11025, 12000, 16000, 22050, 24000, 32000, 44100, 48000]
    for _ in range(int(os.environ.get("WMIX_FUZZ_CASES", "150"))):
        ic, oc, ifr, ofr = int(rng.integers(1, 3)), int(rng.integers(1, 3)), int(rng.choice(rates)), int(rng.choice(rates))
        x = rng.integers(-32768, 32768, size=int(rng.integers(1, 600)) * ic, dtype=np.int16)
        want = np.frombuffer(L.ref_mix("zoom", ic, ifr, oc, ofr, stdin=x.tobytes()), dtype=np.int16)
        assert np.array_equal(orc_zoom(oracle_port, ic, ifr, x, oc, ofr), want), (ic, ifr, oc, ofr, x.size)
        chn, freq = int(rng.integers(1, 3)), int(rng.choice(rates))
        rmode, rarg, nsrc = int(rng.choice([1, 1, 2, 4])), int(rng.choice([1, 1, 2, 4])), int(rng.integers(1, 6))
        sbytes = int(rng.integers(1, 400)) * chn * 2
        start = int(rng.choice([0, 15998, int(rng.integers(0, 8000)) * 2]))
        src = rng.integers(-30000, 30000, size=nsrc * sbytes // 2 + 8, dtype=np.int16)
        b = L.ref_mix("load", freq, chn, rmode, rarg, nsrc, sbytes, start, stdin=src.tobytes())
        ring, meta = orc_load(oracle_port, 1, 8000, freq, chn, rmode, rarg, nsrc, sbytes, start, src)
        assert np.array_equal(ring, np.frombuffer(b[:16000], dtype=np.int16)), (freq, chn, rmode, rarg, nsrc, sbytes, start)
        assert np.array_equal(meta, np.frombuffer(b[16000:], dtype=np.uint32).reshape(nsrc, 2))
