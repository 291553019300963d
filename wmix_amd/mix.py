"""Host-side mirror of wmix's resample + mix functions (src/wmix.h:40-49, 113-127) for batches.
All sample arithmetic happens in wmix_amd/csrc/mix.hip; the cursor walks are index-only host loops."""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib

NULL_HEAD = 0xFFFFFFFF


def len_of_out(in_chn, in_freq, in_len, out_chn, out_freq):
    return lib().wmix_len_of_out(in_chn, in_freq, in_len, out_chn, out_freq)


def len_of_in(in_chn, in_freq, out_chn, out_freq, out_len):
    return lib().wmix_len_of_in(in_chn, in_freq, out_chn, out_freq, out_len)


def pcm_zoom(in_chn, in_freq, pcm, out_chn, out_freq):
    """pcm: int16 CUDA [n_streams, n_in] -> int16 CUDA [n_streams, n_out] (n_out from the reference's own walk)."""
    assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.dim() == 2 and pcm.stride(1) == 1
    in_len = pcm.shape[1] * 2
    # capacity: frames * rate ratio (+ slack); wmix_len_of_out is unit-agnostic and does not bound the byte count
    n_out = (int(np.ceil(pcm.shape[1] / in_chn * max(out_freq / in_freq, 1.0))) + 4) * out_chn
    out = torch.zeros(pcm.shape[0], n_out, dtype=torch.int16, device=pcm.device)
    got = C.c_uint32(0)
    check(lib().wmx_pcm_zoom(in_chn, in_freq, pcm.data_ptr(), in_len, out_chn, out_freq, out.data_ptr(), n_out * 2, pcm.stride(0),
                             out.stride(0), pcm.shape[0], C.byref(got), torch.cuda.current_stream().cuda_stream), "wmx_pcm_zoom")
    return out[:, : got.value // 2]


class MixBatch:
    def __init__(self, n_groups, ring_chn=1, ring_freq=8000):
        self._h = C.c_void_p()
        rc = lib().wmx_mix_create(C.byref(self._h), n_groups, ring_chn, ring_freq)
        if rc != 0:
            self._h = None
            check(rc, "wmx_mix_create")
        self.n_groups = n_groups
        self.ring_bytes = lib().wmx_mix_ring_bytes(self._h)

    def set_play_correct(self, n_bytes):
        """VIEW_PLAY_CORRECT of the reference's platform build (platform/<name>/plat.h): alsa 200 ms of ring (the default), hi3516 / t31 0"""
        check(lib().wmx_mix_set_play_correct(self._h, n_bytes), "wmx_mix_set_play_correct")

    def set(self, head_off=0, tick=0, reduce_mode=1):
        check(lib().wmx_mix_set(self._h, head_off, tick, reduce_mode), "wmx_mix_set")

    def load(self, src, src_bytes, freq, channels, head=NULL_HEAD, tick=0, reduce=1, sample=16):
        """src int16 CUDA [n_groups, n_src, >= src_bytes/2 + channels]; returns (head, tick) after the call."""
        assert src.is_cuda and src.dtype == torch.int16 and src.dim() == 3 and src.stride(2) == 1 and src.shape[0] == self.n_groups
        h, t = C.c_uint32(head), C.c_uint32(tick)
        check(lib().wmx_mix_load(self._h, src.data_ptr(), src_bytes, freq, channels, sample, src.shape[1], src.stride(0), src.stride(1),
                                 reduce, C.byref(h), C.byref(t), torch.cuda.current_stream().cuda_stream), "wmx_mix_load")
        return h.value, t.value

    def drain(self, n_bytes):
        out = torch.empty(self.n_groups, n_bytes // 2, dtype=torch.int16, device="cuda")
        check(lib().wmx_mix_drain(self._h, out.data_ptr(), n_bytes, out.stride(0), torch.cuda.current_stream().cuda_stream), "wmx_mix_drain")
        return out

    def export(self, group=0):
        ring = np.zeros(self.ring_bytes // 2, np.int16)
        h, t = C.c_uint32(0), C.c_uint32(0)
        check(lib().wmx_mix_export(self._h, group, ring.ctypes.data, C.byref(h), C.byref(t)), "wmx_mix_export")
        return ring, h.value, t.value

    def close(self):
        if self._h:
            lib().wmx_mix_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
