"""Host-side mirror of the G.711 entry points over torch device tensors.

Mirrors src/g711codec.h:24-34: encode int16 PCM -> 8-bit codes, decode back.
All arithmetic happens in wmix_amd/csrc/g711.hip.
"""
import torch

from ._lib import check, lib

LAW = {"a": 0, "A": 0, "alaw": 0, "u": 1, "U": 1, "ulaw": 1, 0: 0, 1: 1}


def _stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def encode(law, pcm, out=None):
    """pcm: int16 CUDA tensor (any shape, contiguous) -> uint8 tensor of the same shape."""
    assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.is_contiguous()
    if out is None:
        out = torch.empty(pcm.shape, dtype=torch.uint8, device=pcm.device)
    assert out.is_cuda and out.dtype == torch.uint8 and out.is_contiguous() and out.numel() == pcm.numel()
    check(lib().wmx_g711_encode(LAW[law], pcm.data_ptr(), out.data_ptr(), pcm.numel(), _stream_ptr()), "wmx_g711_encode")
    return out


def decode(law, codes, out=None):
    """codes: uint8 CUDA tensor -> int16 tensor of the same shape."""
    assert codes.is_cuda and codes.dtype == torch.uint8 and codes.is_contiguous()
    if out is None:
        out = torch.empty(codes.shape, dtype=torch.int16, device=codes.device)
    assert out.is_cuda and out.dtype == torch.int16 and out.is_contiguous() and out.numel() == codes.numel()
    check(lib().wmx_g711_decode(LAW[law], codes.data_ptr(), out.data_ptr(), codes.numel(), _stream_ptr()), "wmx_g711_decode")
    return out
