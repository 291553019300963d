"""GPU: the cross-lane FFT executors of the kernels, stand-alone (wmx_debug_fft), bit for bit against the known answers
of the real reference transforms: WebRtc_rdft n = 128 / 256 and aec_rdft_forward/inverse_128 (tests/golden/fft_golden.npz,
SURVEY rows a5 / a13), WebRtcSpl_RealForwardFFT / RealInverseFFT orders 7 / 8 (tests/golden/nsx_golden.npz)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(GOLDEN, "fft_golden.npz"))
X = np.load(os.path.join(GOLDEN, "nsx_golden.npz"))


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def run(wmx, cuda, kind, x, aux=False):
    d = torch.from_numpy(np.ascontiguousarray(x)).to(cuda)
    a = torch.zeros(x.shape[0], dtype=torch.int32, device=cuda)
    assert wmx.wmx_debug_fft(kind, x.shape[0], d.data_ptr(), a.data_ptr(), None) == 0
    return (d.cpu().numpy(), a.cpu().numpy()) if aux else d.cpu().numpy()


def same_floats(a, b):
    """bit-identical, except that a zero may carry either sign (products with the executors' 0 / 1 table entries)"""
    ba, bb = bits(a), bits(b)
    return bool(np.all((ba == bb) | ((a == 0) & (b == 0))))


@pytest.mark.parametrize("kind,n,key", [(0, 128, "ooura_fwd_128"), (1, 128, "ooura_inv_128"), (2, 256, "ooura_fwd_256"),
                                        (3, 256, "ooura_inv_256"), (4, 128, "aec_fwd_128"), (5, 128, "aec_inv_128"),
                                        (6, 128, "aec_fwd_128"), (7, 128, "aec_inv_128")])
def test_float_executors_known_answers(wmx, cuda, kind, n, key):
    got = run(wmx, cuda, kind, G["in_%d" % n])
    assert same_floats(got, G[key])


def test_register_executors_on_more_vectors(wmx, cuda):
    """37 transforms (not a multiple of the 4 per wave of the register form): LDS, register and lane executors agree."""
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((37, 128)) * rng.uniform(1, 3e4, (37, 1))).astype(np.float32)
    assert same_floats(run(wmx, cuda, 6, x), run(wmx, cuda, 4, x))
    assert same_floats(run(wmx, cuda, 7, x), run(wmx, cuda, 5, x))


@pytest.mark.parametrize("order", [7, 8])
def test_spl_fixed_point_fft_known_answers(wmx, cuda, order):
    n = 1 << order
    x = np.zeros((12, n + 2), np.int16)
    x[:, :n] = X["fft_in_%d" % order]
    got = run(wmx, cuda, 8 if order == 7 else 10, x)
    assert np.array_equal(got, X["fft_fwd_%d" % order])
    got, sc = run(wmx, cuda, 9 if order == 7 else 11, X["ifft_in_%d" % order].copy(), aux=True)
    assert np.array_equal(got[:, :n], X["ifft_out_%d" % order]) and np.array_equal(sc, X["ifft_scale_%d" % order])
