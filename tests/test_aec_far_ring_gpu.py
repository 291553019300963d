"""The far-end ring at its limits (round-5 VERDICT next 2).  The reference keeps 250 partitions of far-end spectra per handle
(kBufSizePartitions, W:aec_core.c:37, 1394-1410; 260 KB); this library keeps the 250 SLOTS -- the ring's capacity semantics live in
the host control plane (aec_ctl.h) -- but stores a slot as the partition's 128 int16 samples and makes the two transforms when the
block is consumed (aec.hip AecFarBufs: 64 KB).  Driven here through everything that ring can do, call by call against the REAL
aec_init / aec_setFrameFar / aec_process / aec_process2 of oracle/_ref/libwmixref.so:

  * the normal heartbeat through the start-up phase;
  * aec_setFrameFar alone, 130 packets: the ring fills to its 250 blocks and past them (the oldest unread block is flushed per
    new one, W:aec_core.c:1693-1695);
  * aec_process alone until the ring runs dry and the read pointer is moved BACK into consumed blocks, again and again
    (W:aec_core.c:1788-1793), including -- on a fresh handle -- into slots that were never written (the zeros of WebRtc_InitBuffer);
  * reported delays that jump (0 -> 380 -> 40 -> 500 ms): knownDelay follows after 25 calls and the read pointer moves by
    a hundred blocks in one go, in both directions (W:echo_cancellation.c:821-872, aec_core.c:1795-1802).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from wmix_amd import synth

pytestmark = pytest.mark.gpu


def _schedule(seed, freq):
    """[(kind, reported delay)] per packet: 3 = aec_process2, 1 = aec_setFrameFar, 2 = aec_process"""
    rng = np.random.default_rng(seed)
    s = [(3, 0)] * 60
    s += [(1, 0)] * 130          # far only: 130 x 2.5 (1.25 at 8 kHz) blocks into a ring of 250
    s += [(2, 0)] * 160          # near only: drains it, then the read pointer goes back for more, every sub-frame
    s += [(3, 0)] * 40
    for d in (380, 40, 500, 0):  # the delay filter needs 25 calls to follow
        s += [(3, d)] * 45
    s += [(1, 0)] * 30 + [(2, 120)] * 50 + [(3, 120)] * 30
    s += [(int(rng.integers(1, 4)), int(rng.integers(0, 4)) * 90) for _ in range(120)]  # and anything else, in any order
    return s


def _reference(ref, chn, freq, ims, sched, far, near):
    """one handle of the real wrapper, call by call"""
    vp = C.c_void_p
    ref.aec_init.restype = vp
    ref.aec_init.argtypes = [C.c_int, C.c_int, C.c_int, vp]
    for n, a in (("aec_setFrameFar", [vp, vp, C.c_int]), ("aec_process", [vp, vp, vp, C.c_int, C.c_int]), ("aec_process2", [vp, vp, vp, vp, C.c_int, C.c_int])):
        getattr(ref, n).restype = C.c_int
        getattr(ref, n).argtypes = a
    ref.aec_release.argtypes = [vp]
    ref.aec_release.restype = None
    h = ref.aec_init(chn, freq, ims, None)
    assert h
    pkt = far.shape[1]
    out = near.copy()
    rcs = []
    for k, (kind, d) in enumerate(sched):
        f, o = far[k].ctypes.data, out[k].ctypes.data
        if kind == 1:
            rcs.append(ref.aec_setFrameFar(h, f, pkt // chn))
        elif kind == 2:
            rcs.append(ref.aec_process(h, o, o, pkt // chn, d))
        else:
            rcs.append(ref.aec_process2(h, f, o, o, pkt // chn, d))
    ref.aec_release(h)
    return out, rcs


@pytest.mark.parametrize("freq,ims,fresh", [(16000, 10, False), (8000, 10, False), (8000, 20, False), (16000, 10, True)])
def test_far_ring_full_dry_and_moved_vs_the_real_wrapper(cuda, oracle_ref, freq, ims, fresh):
    from wmix_amd.aec import AecBatch
    sched = _schedule(freq + ims, freq)
    if fresh:  # near-only calls on a handle that has buffered almost nothing: the read pointer goes back into unwritten slots
        sched = [(1, 0)] * 2 + [(2, 0)] * 90 + sched[:200]
    n = len(sched)
    pkt = freq // 1000 * (20 if (ims == 20 and freq == 8000) else 10)
    S = 3
    far = synth.far_end(9800 + freq, n, pkt).reshape(n, pkt)
    near = synth.near_end(9801 + freq, S, n, pkt, far=far.reshape(-1)).reshape(S, n, pkt)
    want = [_reference(oracle_ref, 1, freq, ims, sched, far, np.ascontiguousarray(near[s])) for s in range(S)]
    a = AecBatch(S, 1, freq, ims)
    dfar = torch.from_numpy(far).to(cuda)
    d = torch.from_numpy(near.copy()).to(cuda)
    rcs = []
    for k, (kind, delay) in enumerate(sched):
        if kind == 1:
            rcs.append(a.set_frame_far(dfar[k:k + 1]))
        elif kind == 2:
            rcs.append(a.process(d[:, k:k + 1], delay_ms=delay)[0])
        else:
            rcs.append(a.process2(dfar[k:k + 1], d[:, k:k + 1], delay_ms=delay)[0])
    got = d.cpu().numpy()
    a.close()
    assert rcs == want[0][1]
    for s in range(S):
        assert np.array_equal(got[s], want[s][0]), (freq, ims, fresh, s, int(np.argmax((got[s] != want[s][0]).any(axis=1))))


def test_far_end_footprint(cuda):
    """a far-end costs at most 128 KB of device memory (it was 380 KB): the cohort blob is control plane + slab"""
    from wmix_amd.aec import AecBatch
    a = AecBatch(4, 1, 16000, 10, n_cohorts=2)
    assert a.export_cohort(0).size <= 128 * 1024
    a.close()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    n = 4096
    b = AecBatch(n, 1, 16000, 10, stream_far=np.arange(n))  # every stream its own far-end
    torch.cuda.synchronize()
    used = free0 - torch.cuda.mem_get_info()[0]
    b.close()
    per = used / n
    assert per <= 128 * 1024 + 11232 + 4096, per  # the far-end's slab + the stream's own 11 232 B of state (+ plans, maps)
