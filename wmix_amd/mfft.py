"""Host-side mirror of the reference's stand-alone FFT helpers (math/fft.h:19-51) over torch device tensors.

`fft / fftr / ifft / ifftr` take float32 CUDA tensors shaped [n_batch, N] (or None, which the reference reads as a
NULL array) and return the outputs asked for in `want` ("r", "i", "a" = amplitude curve, "p" = phase curve), each
[n_batch, N].  All arithmetic happens in wmix_amd/csrc/mfft.hip.
"""
import torch

from ._lib import check, lib

KINDS = {"fft": 0, "fftr": 1, "ifft": 2, "ifftr": 3}


def _ptr(t):
    return None if t is None else t.data_ptr()


def transform(kind, re, im=None, want="riap", n=None, n_batch=None):
    k = KINDS[kind] if isinstance(kind, str) else int(kind)
    ref = re if re is not None else im
    if ref is not None:
        assert ref.is_cuda and ref.dtype == torch.float32 and ref.dim() == 2 and ref.is_contiguous()
        n_batch, n = ref.shape
        dev = ref.device
    else:
        dev = torch.device("cuda:0")
    for t in (re, im):
        assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (n_batch, n))
    out = {c: torch.empty((n_batch, n), dtype=torch.float32, device=dev) for c in want if not (k >= 2 and c in "ap")}
    check(lib().wmx_mfft(k, n_batch, n, _ptr(re), _ptr(im), _ptr(out.get("r")), _ptr(out.get("i")), _ptr(out.get("a")),
                         _ptr(out.get("p")), torch.cuda.current_stream().cuda_stream), "wmx_mfft")
    return out


def fft(re, im=None, want="riap"):
    return transform(0, re, im, want)


def fftr(re, want="riap"):
    return transform(1, re, None, want)


def ifft(re, im=None, want="ri"):
    return transform(2, re, im, want)


def ifftr(re, want="ri"):
    return transform(3, re, None, want)


class FftStreams:
    """n independent fft_stream pools (math/fft.c:413-424) of `st_len` samples, resident on the device."""

    def __init__(self, n_streams, st_len, device="cuda:0"):
        self.n, self.st_len = n_streams, st_len
        self.pool = torch.zeros((n_streams, st_len), dtype=torch.float32, device=device)
        self.af = torch.empty_like(self.pool)
        self.pf = torch.empty_like(self.pool)

    def push(self, chunk):
        """chunk: float32 [n_streams, in_len] -> (amplitude, phase) curves [n_streams, st_len] (views, overwritten
        by the next push)."""
        assert chunk.is_cuda and chunk.dtype == torch.float32 and chunk.is_contiguous() and chunk.shape[0] == self.n
        check(lib().wmx_mfft_stream(self.n, chunk.data_ptr(), chunk.shape[1], self.pool.data_ptr(), self.st_len,
                                    self.af.data_ptr(), self.pf.data_ptr(), torch.cuda.current_stream().cuda_stream),
              "wmx_mfft_stream")
        return self.af, self.pf
