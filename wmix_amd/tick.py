"""Host-side mirror of the daemon's tick (include/wmix_amd.h "the daemon's tick"): the play thread's package with the record
heartbeat inside it (src/wmix.c:1347-1440, 528-780) for many mixers side by side.  All sequencing happens in
wmix_amd/csrc/tick.hip; this file only passes device pointers."""
import ctypes as C

import torch

from ._lib import check, lib
from .chain import AEC, AGC, NS, VAD

NULL_HEAD = 0xFFFFFFFF


class TickBatch:
    # the reference's platform directories (platform/<name>/plat.h): (PLAT_AEC_INTERVALMS, PLAT_PLAY_CORRECT in bytes of a 1 x 8000 ring)
    PLATFORMS = {"alsa": (400, 3200), "hi3516": (700, 0), "t31": (0, 0)}

    @classmethod
    def for_platform(cls, platform, n_groups, rec_per_group=1, agc_value=5, stages=NS | AEC | AGC | VAD):
        """the daemon as built from platform/<platform>: 1 x 8000 Hz, 20 ms, that header's echo delay and play-head lead"""
        aec_ms, correct = cls.PLATFORMS[platform]
        tb = cls(n_groups, rec_per_group, 1, 8000, 20, aec_ms, agc_value, stages)
        check(lib().wmx_tick_set_play_correct(tb._h, correct), "wmx_tick_set_play_correct")
        return tb

    # what the reference SHIPS switched on (src/wmix.c:1580-1584: webrtcEnable[WR_NS] = 1, [WR_AGC] = 1, [WR_VAD] = 0, [WR_AEC] = 0);
    # the message thread turns the others on at run time (:1010-1050) -> set_stages
    SHIPPED_STAGES = NS | AGC

    def __init__(self, n_groups, rec_per_group=1, chn=1, freq=8000, interval_ms=20, aec_delay_ms=400, agc_value=5,
                 stages=NS | AEC | AGC | VAD):
        """format defaults = the shipped platform (platform/alsa): 1 x 8000 Hz, WMIX_INTERVAL_MS 20, AEC_INTERVALMS 400, volumeAgc 5.
        `stages` defaults to ALL FOUR switches on (the heartbeat at full length, what the tests and the bench exercise) -- NOT the
        daemon's start-up state, which is SHIPPED_STAGES; stages = 0 is a pure mix / FIFO / zoom tick."""
        self._h = C.c_void_p()
        rc = lib().wmx_tick_create(C.byref(self._h), n_groups, rec_per_group, chn, freq, interval_ms, aec_delay_ms, agc_value, stages)
        if rc != 0:
            self._h = None
            check(rc, "wmx_tick_create")
        self.n_groups, self.rec_per_group, self.chn, self.freq = n_groups, rec_per_group, chn, freq
        self.pkg = lib().wmx_tick_package_samples(self._h)
        self._far = None
        self.head, self.tick = NULL_HEAD, 0  # the sources' common cursor (they began together): NULL head = "first call"

    def load(self, src, src_bytes, freq, channels, reduce=1, sample=16):
        """wmix_load_data of every source of every group for this tick: src int16 CUDA [n_groups, n_src, >= src_bytes / 2 (+ one
        frame of look-ahead when up-sampling)]; the cursor is kept here like a task thread keeps its head / tick."""
        assert src.is_cuda and src.dtype == torch.int16 and src.dim() == 3 and src.stride(2) == 1 and src.shape[0] == self.n_groups
        h, t = C.c_uint32(self.head), C.c_uint32(self.tick)
        check(lib().wmx_tick_load(self._h, src.data_ptr(), src_bytes, freq, channels, sample, src.shape[1], src.stride(0), src.stride(1), reduce,
                                  C.byref(h), C.byref(t), torch.cuda.current_stream().cuda_stream), "wmx_tick_load")
        self.head, self.tick = h.value, t.value

    def set_stages(self, stages, agc_value=-1):
        """webrtcEnable[] at run time (src/wmix.c:1010-1050): a stage that goes is released, one that comes on starts with fresh handles
        (an AGC with agc_value = the daemon's volumeAgc of that moment; < 0: the tick's own)."""
        check(lib().wmx_tick_set_stages(self._h, int(stages), int(agc_value)), "wmx_tick_set_stages")

    def play_ns(self, on=True):
        """webrtcEnable[WR_NS_PA]: ns_process over the played package in front of playPkgBuff_add (src/wmix.c:1370-1386)."""
        check(lib().wmx_tick_play_ns(self._h, 1 if on else 0), "wmx_tick_play_ns")

    def rw_test(self, on=True):
        """wmix->rwTest: the heartbeat loads what it recorded back into the play ring (src/wmix.c:714-732)."""
        check(lib().wmx_tick_rw_test(self._h, 1 if on else 0), "wmx_tick_rw_test")

    def play(self, play=None):
        """The play side of one package; returns the groups' far-end packages (a VIEW of the handle's own [n_groups, pkg] rows:
        valid until the next play)."""
        pp, ps = (play.data_ptr(), play.stride(0)) if play is not None else (None, 0)
        check(lib().wmx_tick_play(self._h, pp, ps, torch.cuda.current_stream().cuda_stream), "wmx_tick_play")
        return self.far()

    def far(self):
        if self._far is None:
            import numpy as np
            ptr = lib().wmx_tick_far(self._h)
            # a tensor over the library's buffer (no copy, not owned): through the CUDA array interface
            holder = type("FarRows", (), {"__cuda_array_interface__": {"shape": (self.n_groups, self.pkg), "typestr": "<i2",
                                                                          "data": (ptr, False), "version": 2, "strides": None}})()
            self._far = torch.as_tensor(holder, device="cuda")
        return self._far

    def record(self, rec, rec_1x8000=None):
        assert rec.is_cuda and rec.dtype == torch.int16 and rec.shape == (self.n_groups * self.rec_per_group, self.pkg) and rec.stride(1) == 1
        got = C.c_uint32(0)
        zp, zs, zc = (rec_1x8000.data_ptr(), rec_1x8000.stride(0), rec_1x8000.shape[1] * 2) if rec_1x8000 is not None else (None, 0, 0)
        check(lib().wmx_tick_record(self._h, rec.data_ptr(), rec.stride(0), zp, zs, zc, C.byref(got), torch.cuda.current_stream().cuda_stream),
              "wmx_tick_record")
        return got.value

    def run(self, rec, play=None, rec_1x8000=None):
        """One package: rec int16 CUDA [n_groups * rec_per_group, pkg] in place; play [n_groups, pkg] and rec_1x8000
        [n_streams, 8000 / 1000 * interval] are filled when given.  Returns the bytes per row written to rec_1x8000."""
        assert rec.is_cuda and rec.dtype == torch.int16 and rec.shape == (self.n_groups * self.rec_per_group, self.pkg) and rec.stride(1) == 1
        got = C.c_uint32(0)
        pp, ps = (play.data_ptr(), play.stride(0)) if play is not None else (None, 0)
        zp, zs, zc = (rec_1x8000.data_ptr(), rec_1x8000.stride(0), rec_1x8000.shape[1] * 2) if rec_1x8000 is not None else (None, 0, 0)
        check(lib().wmx_tick_run(self._h, pp, ps, rec.data_ptr(), rec.stride(0), zp, zs, zc, C.byref(got), torch.cuda.current_stream().cuda_stream),
              "wmx_tick_run")
        return got.value

    def chain_handle(self):
        return lib().wmx_tick_chain(self._h)

    def close(self):
        if self._h:
            lib().wmx_tick_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
