"""math/fft.c on the GPU (wmix_amd/csrc/mfft.hip through the C-ABI) against the golden vectors of the real
reference and against the oracle restatement.  re / im / amplitude: bit-exact.  phase: the device's double atan2
rounded to float may differ from glibc's by one float ulp (tolerance written below)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
from make_mfft_golden import SIZES, mfft_input  # noqa: E402
from oracle import loader  # noqa: E402
from wmix_amd import mfft  # noqa: E402

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(GOLDEN, "mfft_golden.npz"))


def same_bits(a, b):
    return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


def phase_close(a, b):
    """<= 1 float ulp, or both +-pi-adjacent / zero-signed variants of the same angle"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
    return bool(np.all((ulp <= 1) | (a == b)))


def check_outputs(got, want_of, tag):
    for k, v in got.items():
        w = want_of(k)
        g = v.cpu().numpy().reshape(w.shape)
        if k == "p":
            assert phase_close(g, w), (tag, k)
        else:
            assert same_bits(g, w), (tag, k)


@pytest.mark.parametrize("kind", range(4))
@pytest.mark.parametrize("n", SIZES)
def test_golden(cuda, kind, n):
    re, im = mfft_input(n, 7000 + n)
    dre, dim = torch.from_numpy(re[None]).to(cuda), torch.from_numpy(im[None]).to(cuda)
    got = mfft.transform(kind, dre, dim)
    check_outputs(got, lambda k: G["k%d_n%d_%s" % (kind, n, k)], (kind, n))


@pytest.mark.parametrize("kind", range(4))
def test_null_arguments(cuda, kind):
    re, _ = mfft_input(256, 7777)
    got = mfft.transform(kind, torch.from_numpy(re[None]).to(cuda), None)
    check_outputs(got, lambda k: G["k%d_noim_%s" % (kind, k)], kind)
    z = mfft.transform(kind, None, None, want="r", n=64, n_batch=3)
    assert not z["r"].cpu().numpy().any()


def test_stream_golden(cuda):
    sig, _ = mfft_input(160 * 12, 7100)
    st = mfft.FftStreams(1, 1024, cuda)
    for c, chunk in enumerate(sig.reshape(12, 160)):
        af, pf = st.push(torch.from_numpy(chunk[None].copy()).to(cuda))
        assert same_bits(af.cpu().numpy()[0], G["stream_af"][c]), c
        assert phase_close(pf.cpu().numpy()[0], G["stream_pf"][c]), c
    assert same_bits(st.pool.cpu().numpy()[0], G["stream_final"])


@pytest.mark.parametrize("kind,n,batch", [(0, 1024, 257), (1, 1024, 130), (2, 512, 64), (3, 2048, 33), (0, 4096, 9), (1, 8, 1000)])
def test_batches_vs_oracle(cuda, oracle_port, kind, n, batch):
    rng = np.random.default_rng(100 * kind + n)
    re = (rng.standard_normal((batch, n)) * 3000).astype(np.float32)
    im = (rng.standard_normal((batch, n)) * 3000).astype(np.float32)
    got = mfft.transform(kind, torch.from_numpy(re).to(cuda), torch.from_numpy(im).to(cuda))
    got = {k: v.cpu().numpy() for k, v in got.items()}
    for b in range(0, batch, max(1, batch // 16)):
        want = loader.mfft(oracle_port, kind, re[b], im[b], n, prefix="orc")
        for k, w in want.items():
            assert (phase_close if k == "p" else same_bits)(got[k][b], w), (kind, n, b, k)


def test_full_size_properties(cuda):
    """16384 transforms of 1024 points: equal inputs give equal outputs wherever they sit in the batch; the real
    transform's spectrum is conjugate symmetric; the batch agrees with the oracle on a sample of rows."""
    n, batch = 1024, 16384
    base, _ = mfft_input(n, 31)
    x = torch.from_numpy(np.tile(base, (batch, 1))).to(cuda)
    x[1::2] *= 0.5
    o = mfft.fftr(x, want="ri")
    r, i = o["r"], o["i"]
    assert torch.equal(r[0::2], r[0:1].expand(batch // 2, n)) and torch.equal(i[1::2], i[1:2].expand(batch // 2, n))
    assert torch.equal(r[:, 1:n // 2], r[:, n // 2 + 1:].flip(1)) and torch.equal(i[:, 1:n // 2], -i[:, n // 2 + 1:].flip(1))


def test_rejects_bad_sizes(wmx):
    assert wmx.wmx_mfft(0, 1, 48, None, None, None, None, None, None, None) == -10001
    assert wmx.wmx_mfft(0, 1, 8192, None, None, None, None, None, None, None) == -10001
    assert wmx.wmx_mfft(7, 1, 64, None, None, None, None, None, None, None) == -10001
    assert wmx.wmx_mfft_stream(1, None, 16, None, 64, None, None, None) == -10001


@pytest.mark.parametrize("n", [16, 64, 256, 4096])
def test_reference_host_signatures(wmx, oracle_port, n):
    """The legacy math/fft.h functions over host arrays (batch of one): the LDS kernel (16), several transforms per wave with
    one of them real (64), one wave (256), four waves (4096)."""
    re, im = mfft_input(n, 99)
    f4 = lambda: np.zeros(n, np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for kind, name in enumerate(("FFT", "FFTR", "IFFT", "IFFTR")):
        fn = getattr(wmx, name)
        fn.restype = None
        o_r, o_i, o_a, o_p = f4(), f4(), f4(), f4()
        if kind < 2:
            fn.argtypes = [C.c_void_p] * 6 + [C.c_uint]
            fn(p(re), p(im), p(o_r), p(o_i), p(o_a), None, n)
        else:
            fn.argtypes = [C.c_void_p] * 4 + [C.c_uint]
            fn(p(re), p(im), p(o_r), p(o_i), n)
        want = loader.mfft(oracle_port, kind, re, im, n, prefix="orc")
        assert same_bits(o_r, want["r"]) and same_bits(o_i, want["i"])
        if kind < 2:
            assert same_bits(o_a, want["a"])
    # fft_stream
    wmx.fft_stream.restype = None
    wmx.fft_stream.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p]
    st_len = n if n >= 128 else 512
    pool, af = np.zeros(st_len, np.float32), np.zeros(st_len, np.float32)
    chunks = mfft_input(64 * 5, 5)[0].reshape(5, 64)
    for ch in chunks:
        ch = np.ascontiguousarray(ch)
        wmx.fft_stream(p(ch), 64, p(pool), st_len, p(af), None)
    want_pool, want_af, _ = loader.mfft_stream(oracle_port, chunks, st_len, prefix="orc")
    assert same_bits(pool, want_pool) and same_bits(af, want_af[-1])


# ---------------------------------------------------------------- the register-resident kernel (complex sizes 32 .. 4096)
REGS_CASES = [(kind, n) for kind in range(4) for n in ((32, 64, 128, 256, 512, 1024, 2048, 4096) if kind in (0, 2) else (64, 128, 256, 512, 1024, 2048, 4096))]


@pytest.mark.parametrize("kind,n", REGS_CASES)
def test_more_transforms_than_waves_vs_oracle(cuda, oracle_port, kind, n):
    """mfft_regs_kernel's workgroups loop over their share of the batch (a grid of at most 256 x 6 workgroups of 32 ... 1
    transforms): enough transforms that every workgroup takes a second group, with a ragged last group; rows spread over the
    batch (the first, the last, the first of the second round) against the oracle, every output the kind has."""
    nc = n if kind in (0, 2) else n // 2
    batch = {32: 6144 * 16 + 70, 64: 6144 * 8 + 37, 128: 6144 * 4 + 21}.get(nc, 6500)
    rng = np.random.default_rng(900 + 10 * kind + n)
    re = (rng.standard_normal((batch, n)) * 2000).astype(np.float32)
    im = (rng.standard_normal((batch, n)) * 2000).astype(np.float32)
    got = mfft.transform(kind, torch.from_numpy(re).to(cuda), torch.from_numpy(im).to(cuda))
    got = {k: v.cpu().numpy() for k, v in got.items()}
    rows = sorted({0, 1, 3, 4, 31, 32, 255, 1535, 1536, 3071, 3072, 5119, 5120, 6143, 6144, 6145, batch - 40, batch - 33, batch - 2, batch - 1})
    for b in rows:
        want = loader.mfft(oracle_port, kind, re[b], im[b], n, prefix="orc")
        for k, w in want.items():
            assert (phase_close if k == "p" else same_bits)(got[k][b], w), (kind, n, b, k)


@pytest.mark.parametrize("kind,n", [(1, 1024), (3, 512), (1, 2048)])
def test_real_input_that_is_not_8_byte_aligned(cuda, oracle_port, kind, n):
    """FFTR / IFFTR read sample pairs with 8-byte loads when the array allows it; an array that starts on an odd float goes
    through the LDS kernel: same bits."""
    batch = 5
    rng = np.random.default_rng(n + kind)
    flat = torch.from_numpy((rng.standard_normal(batch * n + 1) * 500).astype(np.float32)).to(cuda)
    x = flat[1:].view(batch, n)
    assert x.data_ptr() % 8 == 4 and x.is_contiguous()
    got = mfft.transform(kind, x, None, want="ri")
    aligned = mfft.transform(kind, x.clone(), None, want="ri")
    for k in "ri":
        assert torch.equal(got[k].view(torch.int32), aligned[k].view(torch.int32)), (kind, n, k)
    want = loader.mfft(oracle_port, kind, x[batch - 1].cpu().numpy(), None, n, prefix="orc")
    assert same_bits(got["r"][batch - 1].cpu().numpy(), want["r"]) and same_bits(got["i"][batch - 1].cpu().numpy(), want["i"])


@pytest.mark.parametrize("st_len,in_len", [(32, 16), (64, 20), (128, 64), (256, 100), (512, 256), (1024, 160), (1024, 512), (1024, 1), (2048, 700), (4096, 2048)])
def test_many_streams_vs_oracle(cuda, oracle_port, st_len, in_len):
    """fft_stream on 5 300 (49 189 of the short ones) pools at once, four pushes; pools and curves of a few streams against
    the oracle's fft_stream, all streams of equal input equal."""
    n_streams, pushes = (49189 if st_len <= 128 else 5300), 4  # more pools than one round of the grid takes
    rng = np.random.default_rng(st_len + in_len)
    base = (rng.standard_normal((8, pushes, in_len)) * 1000).astype(np.float32)
    sig = np.ascontiguousarray(base[np.arange(n_streams) % 8])  # [stream, push, in_len]
    st = mfft.FftStreams(n_streams, st_len, cuda)
    for c in range(pushes):
        af, pf = st.push(torch.from_numpy(np.ascontiguousarray(sig[:, c])).to(cuda))
    pool, af, pf = st.pool.cpu().numpy(), af.cpu().numpy(), pf.cpu().numpy()
    for s in range(8):
        want_pool, want_af, want_pf = loader.mfft_stream(oracle_port, base[s], st_len, prefix="orc")
        for row in (s, s + 8 * 640, n_streams - 1 - (n_streams - 1 - s) % 8):
            assert same_bits(pool[row], want_pool), (s, row)
            assert same_bits(af[row], want_af[-1]), (s, row)
            assert phase_close(pf[row], want_pf[-1]), (s, row)
    assert np.array_equal(pool.view(np.uint32)[8:], pool.view(np.uint32)[:-8])
    assert np.array_equal(af.view(np.uint32)[8:], af.view(np.uint32)[:-8])
