"""math/fft.c: the oracle restatement (oracle/orc_mfft.c) against the committed golden vectors of the real
reference, and against the real reference itself where it was built (this container)."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
from make_mfft_golden import SIZES, mfft_input  # noqa: E402
from oracle import loader  # noqa: E402

G = np.load(os.path.join(GOLDEN, "mfft_golden.npz"))


def same_bits(a, b):
    return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


@pytest.mark.parametrize("kind", range(4))
@pytest.mark.parametrize("n", SIZES)
def test_oracle_matches_golden(oracle_port, kind, n):
    re, im = mfft_input(n, 7000 + n)
    o = loader.mfft(oracle_port, kind, re, im, n, prefix="orc")
    for k, v in o.items():
        assert same_bits(v, G["k%d_n%d_%s" % (kind, n, k)]), (kind, n, k)


@pytest.mark.parametrize("kind", range(4))
def test_oracle_null_imaginary(oracle_port, kind):
    re, _ = mfft_input(256, 7777)
    o = loader.mfft(oracle_port, kind, re, None, 256, prefix="orc")
    for k, v in o.items():
        assert same_bits(v, G["k%d_noim_%s" % (kind, k)]), (kind, k)
    # NULL outputs are skipped, NULL real input reads as zeros (math/fft.c:24-34)
    z = loader.mfft(oracle_port, kind, None, None, 64, prefix="orc", want="r")
    assert not z["r"].any()


def test_oracle_stream(oracle_port):
    sig, _ = mfft_input(160 * 12, 7100)
    stream, afs, pfs = loader.mfft_stream(oracle_port, sig.reshape(12, 160), 1024, prefix="orc")
    assert same_bits(stream, G["stream_final"])
    assert same_bits(np.stack(afs), G["stream_af"]) and same_bits(np.stack(pfs), G["stream_pf"])


def test_rejects_bad_sizes(oracle_port):
    fn = oracle_port.orc_mfft
    assert fn(0, None, None, None, None, None, None, 0) == -1
    assert fn(0, None, None, None, None, None, None, 48) == -1


@pytest.mark.skipif(not loader.have_ref(), reason="real reference not built here")
@pytest.mark.parametrize("n", [2, 8, 64, 1024, 4096])
def test_oracle_against_real_reference(oracle_port, oracle_ref, n):
    rng = np.random.default_rng(n)
    for trial in range(3):
        re = (rng.standard_normal(n) * 10 ** rng.integers(0, 5)).astype(np.float32)
        im = (rng.standard_normal(n) * 10 ** rng.integers(0, 5)).astype(np.float32)
        for kind in range(4):
            a = loader.mfft(oracle_ref, kind, re, im, n)
            b = loader.mfft(oracle_port, kind, re, im, n, prefix="orc")
            for k in a:
                assert same_bits(a[k], b[k]), (n, kind, k)


def test_forward_inverse_round_trip(oracle_port):
    """IFFT(FFT(x)) == x up to float rounding (the inverse's per-stage halving makes it the true inverse)."""
    re, im = mfft_input(512, 1)
    f = loader.mfft(oracle_port, 0, re, im, 512, prefix="orc", want="ri")
    b = loader.mfft(oracle_port, 2, f["r"], f["i"], 512, prefix="orc", want="ri")
    assert np.abs(b["r"] - re).max() < 0.05 and np.abs(b["i"] - im).max() < 0.05
