"""What ONE heartbeat of the unchanged daemon costs through the legacy adapters (include/wmix_compat.h): the record chain of
src/wmix.c:613-709 -- ns_process -> aec_process2 -> agc_process -> vad_process on HOST pointers, handles made with the daemon's
own arguments (WMIX_INTERVAL_MS = 20, src/wmixConf.h:112; 20 ms per heartbeat) -- next to the same heartbeat through the real
reference compiled on this host (oracle/_ref/libwmixref.so), when it is there.  Every adapter call is a batch of ONE stream:
H2D copy -> launch -> D2H copy, synchronous (round-3 VERDICT: DESIGN quoted ~30 us per call without a measurement).

    python tools_dev/legacy_latency.py            one JSON object per (rate, channels) on stdout
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def heartbeat_latency(freq=8000, chn=1, n_beats=600, warm=100, check=True):
    """Returns a dict: microseconds per 20 ms heartbeat through the four legacy adapters (mean, median, p99), per adapter, and --
    when oracle/_ref is present -- through the reference on this host; `parity` = max |LSB| against the oracle restatement."""
    from make_aec_golden import aec_input
    from oracle import loader as L
    from wmix_amd import _lib
    W = _lib.lib()
    per = freq // 50  # frames per 20 ms heartbeat
    n10 = n_beats * 2
    far, near = aec_input(chn, freq, 10, n10, seed=7300 + freq // 8000 + chn)
    buf = near.copy()
    dbg = None
    ns = W.ns_init(chn, freq, dbg)
    aec = W.aec_init(chn, freq, 20, dbg)
    agc = W.agc_init(chn, freq, 20, 5, dbg)
    vad = W.vad_init(chn, freq, 20, dbg)
    assert ns and aec and agc and vad
    step = per * chn
    t_stage = np.zeros((n_beats, 4))
    for b in range(n_beats):
        p = C.c_void_p(buf.ctypes.data + 2 * b * step)
        f = C.c_void_p(far.ctypes.data + 2 * b * step)
        t0 = time.perf_counter()
        W.ns_process(ns, p, p, per)
        t1 = time.perf_counter()
        rc = W.aec_process2(aec, f, p, p, per, 0)
        t2 = time.perf_counter()
        rc2 = W.agc_process(agc, p, p, per)
        t3 = time.perf_counter()
        W.vad_process(vad, p, per)
        t4 = time.perf_counter()
        assert rc == 0 and rc2 == 0
        t_stage[b] = (t1 - t0, t2 - t1, t3 - t2, t4 - t3)
    W.ns_release(ns), W.aec_release(aec), W.agc_release(agc), W.vad_release(vad)
    beat = t_stage[warm:].sum(1) * 1e6
    out = {"freq": freq, "chn": chn, "heartbeats": n_beats - warm, "interval_ms": 20,
           "adapters_us_per_heartbeat": {"mean": float(beat.mean()), "median": float(np.median(beat)), "p99": float(np.percentile(beat, 99))},
           "per_adapter_us_mean": dict(zip(("ns_process", "aec_process2", "agc_process", "vad_process"),
                                           [float(x) for x in t_stage[warm:].mean(0) * 1e6]))}
    if check:
        want = L.run_chain(L.port(), chn, freq, 5, 15, far, near, per, prefix="orc", interval_ms=20)
        out["parity_max_lsb_vs_oracle"] = int(np.abs(buf.astype(np.int32) - want.astype(np.int32)).max())
    if L.have_ref():
        try:
            ref = L.ref()
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                L.run_chain(ref, chn, freq, 5, 15, far, near, per, prefix="ref", interval_ms=20)
            out["reference_cpu_us_per_heartbeat"] = (time.perf_counter() - t0) / (reps * n_beats) * 1e6
        except Exception as e:  # a prebuilt library that does not load here
            out["reference_cpu_us_per_heartbeat"] = None
            out["reference_error"] = str(e)
    return out


def packet_edge_latency(n_calls=600, warm=100):
    """The task threads' per-packet legacy calls on host pointers: PCM2G711a / G711a2PCM of one RTP payload (160 samples,
    src/wmixTask.c:1139, 1282), wmix_pcm_zoom of one 20 ms package (src/wmix.c:730).  Microseconds per call (mean, p99)."""
    from wmix_amd import _lib
    W = _lib.lib()
    rng = np.random.default_rng(3)
    pcm = rng.integers(-20000, 20000, 160, dtype=np.int16)
    codes = np.zeros(160, np.uint8)
    back = np.zeros(160, np.int16)
    big = rng.integers(-20000, 20000, 1280, dtype=np.int16)  # 20 ms of 2 x 16000
    small = np.zeros(1280, np.int16)
    for f in (W.PCM2G711a, W.G711a2PCM):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        f.restype = C.c_int
    W.wmix_pcm_zoom.argtypes = [C.c_uint8, C.c_uint16, C.c_void_p, C.c_uint32, C.c_uint8, C.c_uint16, C.c_void_p]
    W.wmix_pcm_zoom.restype = C.c_uint32
    calls = {"PCM2G711a_160": lambda: W.PCM2G711a(pcm.ctypes.data, codes.ctypes.data, 320, 0),
             "G711a2PCM_160": lambda: W.G711a2PCM(codes.ctypes.data, back.ctypes.data, 160, 0),
             "wmix_pcm_zoom_2x16000_to_1x8000_20ms": lambda: W.wmix_pcm_zoom(2, 16000, big.ctypes.data, 2560, 1, 8000, small.ctypes.data)}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_mix_gpu import Head, Point  # the WMix_Struct head as ctypes (tests/test_mix_gpu.py)
    W.wmix_load_data.restype = Point
    W.wmix_load_data.argtypes = [C.POINTER(Head), Point, C.c_uint32, C.c_uint16, C.c_uint8, C.c_uint8, Point, C.c_uint8, C.POINTER(C.c_uint32)]
    ring = np.zeros(8008, np.int16)
    w = Head()
    w.start.U8, w.end.U8, w.head.U8 = ring.ctypes.data, ring.ctypes.data + 16000, ring.ctypes.data
    w.run, w.reduceMode, w.tick = True, 1, 0
    cur = {"head": None, "tick": C.c_uint32(0)}

    def load():  # one task thread's 20 ms chunk of 2 x 16000 into the 1 x 8000 ring, its cursor carried along
        sp, hp = Point(), Point()
        sp.U8, hp.U8 = big.ctypes.data, cur["head"]
        cur["head"] = W.wmix_load_data(C.byref(w), sp, 2560, 16000, 2, 16, hp, 1, C.byref(cur["tick"])).U8

    calls["wmix_load_data_2x16000_20ms"] = load
    x = rng.standard_normal(1024).astype(np.float32)
    o = [np.zeros(1024, np.float32) for _ in range(4)]
    W.FFTR.argtypes = [C.c_void_p] * 6 + [C.c_uint]
    W.FFTR.restype = None
    calls["FFTR_1024"] = lambda: W.FFTR(x.ctypes.data, None, o[0].ctypes.data, o[1].ctypes.data, o[2].ctypes.data, o[3].ctypes.data, 1024)
    out = {}
    for name, fn in calls.items():
        t = np.zeros(n_calls)
        for k in range(n_calls):
            t0 = time.perf_counter()
            fn()
            t[k] = time.perf_counter() - t0
        out[name] = {"mean_us": float(t[warm:].mean() * 1e6), "p99_us": float(np.percentile(t[warm:], 99) * 1e6)}
    return {"packet_edge_legacy_calls": out}


if __name__ == "__main__":
    for freq, chn in ((8000, 1), (16000, 1), (16000, 2)):
        print(json.dumps(heartbeat_latency(freq, chn)))
    print(json.dumps(packet_edge_latency()))
