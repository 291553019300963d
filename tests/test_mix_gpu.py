"""GPU parity: wmix_amd/csrc/mix.hip through the C ABI vs the goldens of the real reference mixer arithmetic
and vs the oracle for other ring formats / many groups.  Integer path: bit-exact."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
from make_mix_golden import LOAD_CASES, ZOOM_CASES, load_input, zoom_input  # noqa: E402
from test_mix_oracle import _bind, orc_load, orc_zoom  # noqa: E402

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(GOLDEN, "mix_golden.npz"))


def gpu_load(cuda, ring_chn, ring_freq, freq, chn, rmode, rarg, nsrc, sbytes, start, src, n_groups=1, play_correct=None):
    import torch
    from wmix_amd.mix import MixBatch
    mb = MixBatch(n_groups, ring_chn, ring_freq)
    mb.set(start, 0, rmode)
    if play_correct is not None:
        mb.set_play_correct(play_correct)
    per = sbytes // 2
    # sources back to back like the reference driver: source s starts at s*per, look-ahead reads the next one
    d = torch.from_numpy(np.ascontiguousarray(np.tile(src[None, :], (n_groups, 1)))).to(cuda)
    view = torch.as_strided(d, (n_groups, nsrc, per + chn), (d.stride(0), per, 1))
    h, t = mb.load(view, sbytes, freq, chn, reduce=rarg)
    rings = [mb.export(g)[0] for g in range(n_groups)]
    mb.close()
    return rings, h, t


@pytest.mark.parametrize("i", range(len(ZOOM_CASES)))
def test_zoom_golden(cuda, wmx, i):
    import torch
    from wmix_amd import mix
    ic, ifr, oc, ofr, n = ZOOM_CASES[i]
    x = zoom_input(i, n)
    got = mix.pcm_zoom(ic, ifr, torch.from_numpy(np.tile(x, (3, 1))).to(cuda), oc, ofr).cpu().numpy()
    assert got.shape[1] == G["zoom_%d" % i].size and all(np.array_equal(got[s], G["zoom_%d" % i]) for s in range(3))
    assert mix.len_of_out(ic, ifr, n, oc, ofr) == G["lens_%d" % i][0]
    assert mix.len_of_in(ic, ifr, oc, ofr, n) == G["lens_%d" % i][1]
    out = np.zeros(n * 8 + 64, np.uint8)  # legacy host signature, src/wmix.h:122-127
    m = wmx.wmix_pcm_zoom(ic, ifr, x.ctypes.data_as(C.c_void_p), n, oc, ofr, out.ctypes.data_as(C.c_void_p))
    assert m == G["zoom_%d" % i].size * 2 and np.array_equal(out[:m].view(np.int16), G["zoom_%d" % i])


@pytest.mark.parametrize("i", range(len(LOAD_CASES)))
def test_load_data_golden(cuda, i):
    freq, chn, rmode, rarg, nsrc, sbytes, start = LOAD_CASES[i]
    rings, h, t = gpu_load(cuda, 1, 8000, freq, chn, rmode, rarg, nsrc, sbytes, start, load_input(i, nsrc, sbytes), n_groups=3)
    for r in rings:
        assert np.array_equal(r, G["ring_%d" % i])
    assert (t, h) == tuple(G["meta_%d" % i][-1])  # every source of one call ends on the same cursor


@pytest.mark.parametrize("ring_chn,ring_freq", [(2, 16000), (1, 16000), (2, 8000)])
def test_other_ring_formats_vs_oracle(cuda, oracle_port, ring_chn, ring_freq):
    """ring formats other than the reference's default build can only be pinned by the restatement."""
    _bind(oracle_port)
    rng = np.random.default_rng(77)
    for (freq, chn, rmode, rarg, nsrc, sbytes, start) in ((32000, 2, 1, 1, 3, 1280, 0), (8000, 1, 2, 1, 2, 320, 64), (ring_freq, ring_chn, 1, 1, 4, 640, 0),
                                                          (11025, 2, 1, 1, 2, 884, ring_chn * 2 * ring_freq - 128)):
        src = rng.integers(-20000, 20000, size=nsrc * sbytes // 2 + 8, dtype=np.int16)
        want, meta = orc_load(oracle_port, ring_chn, ring_freq, freq, chn, rmode, rarg, nsrc, sbytes, start, src)
        rings, h, t = gpu_load(cuda, ring_chn, ring_freq, freq, chn, rmode, rarg, nsrc, sbytes, start, src)
        assert np.array_equal(rings[0], want) and (t, h) == tuple(meta[-1])


class Point(C.Union):
    _fields_ = [("U8", C.c_void_p)]


class Head(C.Structure):  # WMix_Struct_Head (include/wmix_compat.h)
    _fields_ = [("objAo", C.c_void_p), ("objAi", C.c_void_p), ("buff", C.c_void_p), ("start", Point), ("end", Point), ("head", Point),
                ("tail", Point), ("run", C.c_bool), ("loopWord", C.c_uint8), ("loopWordRecord", C.c_uint8), ("loopWordFifo", C.c_uint8),
                ("loopWordRtp", C.c_uint8), ("tick", C.c_uint32), ("thread_sys", C.c_uint32), ("thread_record", C.c_uint32),
                ("thread_play", C.c_uint32), ("playRun", C.c_bool), ("recordRun", C.c_bool), ("shmemRun", C.c_int), ("msg_key", C.c_int),
                ("msg_fd", C.c_int), ("reduceMode", C.c_uint8)]


def legacy_load(wmx, freq, chn, rmode, rarg, nsrc, sbytes, start, src):
    """nsrc calls of the legacy wmix_load_data (NULL head each) into a fresh 1 x 8000 host ring -> (ring, [(tick, head)])"""
    wmx.wmix_load_data.restype = Point
    wmx.wmix_load_data.argtypes = [C.POINTER(Head), Point, C.c_uint32, C.c_uint16, C.c_uint8, C.c_uint8, Point, C.c_uint8, C.POINTER(C.c_uint32)]
    ring = np.zeros(16000 // 2 + 8, np.int16)
    w = Head()
    w.start.U8 = ring.ctypes.data
    w.end.U8 = ring.ctypes.data + 16000
    w.head.U8 = ring.ctypes.data + start
    w.run, w.reduceMode, w.tick = True, rmode, 0
    meta = []
    for s in range(nsrc):
        tick = C.c_uint32(0)
        sp, hp = Point(), Point()
        sp.U8 = src.ctypes.data + s * sbytes
        hp.U8 = None
        r = wmx.wmix_load_data(C.byref(w), sp, sbytes, freq, chn, 16, hp, rarg, C.byref(tick))
        meta.append((tick.value, r.U8 - ring.ctypes.data if r.U8 else None))
    return ring[:8000].copy(), meta


@pytest.mark.parametrize("platform", ["hi3516", "t31"])
def test_play_correct_of_the_other_platform_builds(cuda, oracle_port, platform):
    """PLAT_PLAY_CORRECT = 0 (platform/hi3516/plat.h:16, platform/t31/plat.h:16): a source without a cursor starts AT the play head
    (src/wmix.c:1668-1669).  wmx_mix_set_play_correct against the restatement and -- where oracle/_ref travelled -- against
    src/wmix.c compiled with that platform's header; then the legacy wmix_load_data with WMIX_AMD_PLAY_CORRECT=0 in a process of its
    own (the adapter reads its environment once)."""
    import subprocess
    from conftest import ROOT
    from oracle import loader as L
    _bind(oracle_port)
    correct = L.PLATFORMS[platform][1]
    rng = np.random.default_rng(99)
    for (freq, chn, rmode, rarg, nsrc, sbytes, start) in ((32000, 2, 1, 1, 3, 1280, 0), (8000, 1, 2, 1, 2, 320, 64), (11025, 2, 1, 1, 2, 884, 15872),
                                                          (44100, 1, 4, 1, 3, 1764, 8000)):
        src = rng.integers(-20000, 20000, size=nsrc * sbytes // 2 + 8, dtype=np.int16)
        want, meta = L.mix_load(oracle_port, 1, 8000, freq, chn, rmode, rarg, nsrc, sbytes, start, src, play_correct=correct)
        rings, h, t = gpu_load(cuda, 1, 8000, freq, chn, rmode, rarg, nsrc, sbytes, start, src, play_correct=correct)
        assert np.array_equal(rings[0], want) and (t, h) == tuple(meta[-1])
        if L.have_ref_mix(platform):
            b = L.ref_mix("load", freq, chn, rmode, rarg, nsrc, sbytes, start, stdin=src.tobytes(), platform=platform)
            assert np.array_equal(rings[0], np.frombuffer(b[:16000], dtype=np.int16))
    code = ("import sys, numpy as np; sys.path[:0] = [%r, %r]; from wmix_amd import _lib; from test_mix_gpu import legacy_load;"
            "src = np.random.default_rng(5).integers(-20000, 20000, size=2000, dtype=np.int16);"
            "ring, meta = legacy_load(_lib.lib(), 32000, 2, 1, 1, 3, 1280, 600, src); np.save(sys.argv[1], ring); print(meta)")
    env = dict(os.environ, WMIX_AMD_PLAY_CORRECT=str(correct))
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "legacy_ring_%s_%d.npy" % (platform, os.getpid()))
    r = subprocess.run([sys.executable, "-c", code % (ROOT, os.path.join(ROOT, "tests")), out], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    src = np.random.default_rng(5).integers(-20000, 20000, size=2000, dtype=np.int16)
    want, meta = L.mix_load(oracle_port, 1, 8000, 32000, 2, 1, 1, 3, 1280, 600, src, play_correct=correct)
    assert np.array_equal(np.load(out), want) and r.stdout.strip().splitlines()[-1] == str([(int(a), int(b)) for a, b in meta])
    os.remove(out)
    bad = subprocess.run([sys.executable, "-c", code % (ROOT, os.path.join(ROOT, "tests")), out], env=dict(env, WMIX_AMD_PLAY_CORRECT="16001"),
                         capture_output=True, text=True, timeout=300)  # not inside the ring: every call refuses, the ring stays empty
    assert bad.returncode == 0 and "WMIX_AMD_PLAY_CORRECT=16001" in bad.stderr and not np.load(out).any()
    os.remove(out)


def test_legacy_wmix_load_data_signature(wmx):
    """WMix_Point wmix_load_data(WMix_Struct*, ...) over a host ring (src/wmix.h:40-49), default 1 x 8000 ring."""
    wmx.wmix_load_data.restype = Point
    wmx.wmix_load_data.argtypes = [C.POINTER(Head), Point, C.c_uint32, C.c_uint16, C.c_uint8, C.c_uint8, Point, C.c_uint8, C.POINTER(C.c_uint32)]
    i = 2
    freq, chn, rmode, rarg, nsrc, sbytes, start = LOAD_CASES[i]
    src = load_input(i, nsrc, sbytes)
    ring = np.zeros(16000 // 2 + 8, np.int16)
    w = Head()
    w.start.U8 = ring.ctypes.data
    w.end.U8 = ring.ctypes.data + 16000
    w.head.U8 = ring.ctypes.data + start
    w.run, w.reduceMode, w.tick = True, rmode, 0
    for s in range(nsrc):
        tick = C.c_uint32(0)
        sp, hp = Point(), Point()
        sp.U8 = src.ctypes.data + s * sbytes
        hp.U8 = None
        r = wmx.wmix_load_data(C.byref(w), sp, sbytes, freq, chn, 16, hp, rarg, C.byref(tick))
        assert (tick.value, r.U8 - ring.ctypes.data) == tuple(G["meta_%d" % i][s])
    assert np.array_equal(ring[:8000], G["ring_%d" % i])


def test_full_size_mix_properties(cuda):
    """configs[4] size: 32768 sources of 2ch 32 kHz 10 ms mixed 8-way into 4096 groups of a 1 x 8000 ring.
    Equal inputs give equal rings; draining returns the mixed packet and leaves the ring all zero; a mix of
    sources that never saturates equals the plain integer sum."""
    import torch
    from wmix_amd.mix import MixBatch
    n_groups, n_src, per = 4096, 8, 640
    base = np.random.default_rng(5).integers(-3000, 3000, size=(2, n_src, per + 2), dtype=np.int16)
    d = torch.from_numpy(base[np.arange(n_groups) % 2]).to(cuda)
    mb = MixBatch(n_groups, 1, 8000)
    h, t = mb.load(d, per * 2, 32000, 2)
    assert (h, t) == (3200 + 160, 3200 + 160)
    r0, r1, r2 = mb.export(0)[0], mb.export(1)[0], mb.export(4094)[0]
    assert np.array_equal(r0, r2) and not np.array_equal(r0, r1)
    want = base[0, :, 0:per:8].astype(np.int32).sum(0)  # L channel, every 4th frame, summed over sources
    assert np.array_equal(r0[1600:1680].astype(np.int32), want)
    mb.set(3200, 0, 1)
    out = mb.drain(160).cpu().numpy()
    assert np.array_equal(out[0], r0[1600:1680]) and np.array_equal(out[1], r1[1600:1680])
    assert not mb.export(0)[0].any()
    mb.close()


def _legacy_types(wmx):
    class Point(C.Union):
        _fields_ = [("U8", C.c_void_p)]

    class Head(C.Structure):  # WMix_Struct_Head (include/wmix_compat.h)
        _fields_ = [("objAo", C.c_void_p), ("objAi", C.c_void_p), ("buff", C.c_void_p), ("start", Point), ("end", Point), ("head", Point),
                    ("tail", Point), ("run", C.c_bool), ("loopWord", C.c_uint8), ("loopWordRecord", C.c_uint8), ("loopWordFifo", C.c_uint8),
                    ("loopWordRtp", C.c_uint8), ("tick", C.c_uint32), ("thread_sys", C.c_uint32), ("thread_record", C.c_uint32),
                    ("thread_play", C.c_uint32), ("playRun", C.c_bool), ("recordRun", C.c_bool), ("shmemRun", C.c_int), ("msg_key", C.c_int),
                    ("msg_fd", C.c_int), ("reduceMode", C.c_uint8)]

    wmx.wmix_load_data.restype = Point
    wmx.wmix_load_data.argtypes = [C.POINTER(Head), Point, C.c_uint32, C.c_uint16, C.c_uint8, C.c_uint8, Point, C.c_uint8, C.POINTER(C.c_uint32)]
    return Point, Head


def test_legacy_load_data_touches_only_its_span(wmx, oracle_port):
    """The reference writes ring bytes [head, head + n_out*2) and nothing else while other threads work on the ring
    (src/wmix.c:1347-1352, 1678-1702).  A second thread keeps rewriting ring samples OUTSIDE the span during many legacy
    calls (ctypes releases the GIL for the call): none of its writes may be lost or resurrected, and the span itself
    must equal the oracle's ring, including across the wrap.  Also: the copy branch must not read behind the source."""
    import threading
    Point, Head = _legacy_types(wmx)
    _bind(oracle_port)
    ring = np.zeros(8000 + 8, np.int16)
    w = Head()
    w.start.U8, w.end.U8 = ring.ctypes.data, ring.ctypes.data + 16000
    start = 16000 - 100  # the span of every call wraps
    w.head.U8 = ring.ctypes.data + start
    w.run, w.reduceMode, w.tick = True, 1, 0
    # source buffer that ends exactly at the end of an allocation guard: the 320 B same-format source sits at the tail
    rng = np.random.default_rng(3)
    src = rng.integers(-9000, 9000, size=160, dtype=np.int16)
    outside = np.r_[200:3000]  # sample indices away from the span [7950, 8000) + [0, 110)
    stop = threading.Event()
    writes = [0]

    def player():  # plays the part of wmix_play_thread zeroing / other tasks adding elsewhere in the ring
        k = 0
        while not stop.is_set():
            k += 1
            ring[outside] = k & 0x7FFF
            writes[0] = k

    th = threading.Thread(target=player)
    th.start()
    try:
        for _ in range(200):
            tick = C.c_uint32(0)
            sp, hp = Point(), Point()
            sp.U8, hp.U8 = src.ctypes.data, ring.ctypes.data + start  # an explicit head: the span wraps at the ring end
            r = wmx.wmix_load_data(C.byref(w), sp, 320, 8000, 1, 16, hp, 1, C.byref(tick))
            assert r.U8 - ring.ctypes.data == (start + 320) % 16000 and tick.value == 320
            assert r.U8 is not None
    finally:
        stop.set()
        th.join()
    # the player's last write is intact everywhere outside the span (a whole-ring write-back would have restored
    # older values somewhere in these 2 900 samples at some point of 200 calls)
    assert (ring[outside] == (writes[0] & 0x7FFF)).all()
    # the span: 200 saturating adds of the same packet at the given head
    pos = start // 2
    want = np.zeros(8000, np.int16)
    acc = np.zeros(160, np.int32)
    for _ in range(200):
        a, b = acc, src.astype(np.int32)
        acc = np.where(a == 0, b, np.where(b == 0, a, np.clip(a + b, -32768, 32767)))
    idx = (pos + np.arange(160)) % 8000
    want[idx] = acc
    got = ring[:8000].copy()
    got[outside] = 0
    assert np.array_equal(got, want)


def test_zoom_capacity_is_checked(cuda, wmx):
    import torch
    x = torch.zeros(2, 160, dtype=torch.int16, device=cuda)
    out = torch.full((2, 400), 77, dtype=torch.int16, device=cuda)
    need = C.c_uint32(0)
    rc = wmx.wmx_pcm_zoom(1, 8000, x.data_ptr(), 320, 1, 16000, out.data_ptr(), 100, 160, 400, 2, C.byref(need), None)
    assert rc == -10001 and need.value == 640
    torch.cuda.synchronize()
    assert (out == 77).all()
    rc = wmx.wmx_pcm_zoom(1, 8000, x.data_ptr(), 320, 1, 16000, out.data_ptr(), 640, 160, 400, 2, C.byref(need), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert (out[:, :320] == 0).all() and (out[:, 320:] == 77).all()


def test_load_rejects_more_than_one_ring(cuda):
    import torch
    from wmix_amd._lib import WmxError
    from wmix_amd.mix import MixBatch
    mb = MixBatch(1, 1, 8000)
    d = torch.zeros(1, 1, 9000, dtype=torch.int16, device=cuda)
    with pytest.raises(WmxError):
        mb.load(d, 17000, 8000, 1)
    mb.close()


def test_variable_chunk_sizes_walk_through_the_schedule_cache(cuda, wmx, oracle_port):
    """Round-2 ADVICE: callers with variable chunk sizes (file tails, RTP) used to hit a device-wide synchronisation and a
    mass eviction every 64 formats.  150 different lengths through wmx_pcm_zoom, twice (the second pass meets evicted
    entries again), every result equal to the oracle's wmix_pcm_zoom."""
    import torch
    _bind(oracle_port)
    o = oracle_port
    rng = np.random.default_rng(77)
    x = rng.integers(-30000, 30000, 2 * 4096, dtype=np.int16)
    d = torch.from_numpy(x).to(cuda)
    out = torch.zeros(8192, dtype=torch.int16, device=cuda)
    for rep in range(2):
        for k in range(150):
            in_len = 2 * (2 * (100 + 7 * k))  # bytes of 2-channel 32 kHz input
            need = C.c_uint32(0)
            out.fill_(123)
            rc = wmx.wmx_pcm_zoom(2, 32000, d.data_ptr(), in_len, 1, 8000, out.data_ptr(), 16384, 0, 0, 1, C.byref(need), None)
            assert rc == 0
            want = orc_zoom(o, 2, 32000, x[: in_len // 2], 1, 8000)
            got = out[: need.value // 2].cpu().numpy()
            assert need.value == want.size * 2 and np.array_equal(got, want), (rep, k)


def test_legacy_adapters_on_short_lived_task_threads(wmx):
    """The daemon starts a thread per play / record task (src/wmixTask.c) and those threads call wmix_pcm_zoom /
    wmix_load_data; their staging buffers and device ring are thread-local and must go back when the thread ends
    (round-2 ADVICE: they were leaked).  200 threads, each converting once: free device memory afterwards is what it was."""
    import threading
    import torch
    x = (np.arange(640) * 37 % 2000 - 1000).astype(np.int16)

    def task(res, i):
        out = np.zeros(640, np.int16)
        n = wmx.wmix_pcm_zoom(2, 32000, x.ctypes.data_as(C.c_void_p), 1280, 1, 8000, out.ctypes.data_as(C.c_void_p))
        res[i] = (n, out[:80].copy())

    res = {}
    t = threading.Thread(target=task, args=(res, -1))
    t.start()
    t.join()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for i in range(200):
        t = threading.Thread(target=task, args=(res, i))
        t.start()
        t.join()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert all(res[i][0] == 160 and np.array_equal(res[i][1], res[-1][1]) for i in range(200))
    assert free0 - free1 < 8 << 20, "device memory shrank by %d bytes over 200 task threads" % (free0 - free1)
