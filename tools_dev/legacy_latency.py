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


if __name__ == "__main__":
    for freq, chn in ((8000, 1), (16000, 1), (16000, 2)):
        print(json.dumps(heartbeat_latency(freq, chn)))
