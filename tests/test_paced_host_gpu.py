"""examples/host_paced.c: the reference's paced heartbeat (src/wmix.c:536-538, 820) for S streams from plain C over wmx_rt_* --
clock_nanosleep(TIMER_ABSTIME) releases, latency = scheduled release -> last row in host memory.  At a small S every tick must
meet the reference's budget (tick - 2 ms) and the rows of the sampled streams, kept for the last ticks, must be the oracle's;
bench.py --paced (the same loop from Python) must agree."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

HOST = os.path.join(ROOT, "examples", "host_paced")


@pytest.mark.parametrize("kind,tick_ms,freq,interval_ms,phases,calls", [("pcm16k", 20, 16000, 20, 1, 0), ("pcm16k", 10, 16000, 10, 1, 0), ("rtp8k", 20, 8000, 20, 1, 0),
                                                                        ("pcm16k", 20, 16000, 20, 3, 0), ("rtp8k", 20, 8000, 20, 4, 0),
                                                                        ("pcm16k", 20, 16000, 20, 2, 1)])  # calls: a far-end per stream
def test_host_paced_meets_the_budget_and_the_oracle(cuda, tmp_path, kind, tick_ms, freq, interval_ms, phases, calls):
    import bench
    assert os.path.exists(HOST), "examples/host_paced is built by __graft_entry__.build()"
    S, sub, slots, ticks, prime, keep, n_pat = 3000, 1024, 4, 120, 90, 16, 64
    far, rows = bench.paced_pattern(kind, slots, interval_ms, n_pattern=n_pat, n_far=16 if calls else 1)
    pat = tmp_path / "pattern.bin"
    with open(pat, "wb") as f:
        f.write(np.ascontiguousarray(far).tobytes())
        f.write(np.ascontiguousarray(rows).tobytes())
    sample = [0, 1, 1023, 1024, 2047, 2048, 2999]
    dump, lat = tmp_path / "dump.bin", tmp_path / "lat.f64"
    cmd = [HOST, "--streams", str(S), "--sub", str(sub), "--slots", str(slots), "--tick-ms", str(tick_ms), "--ticks", str(ticks), "--prime", str(prime),
           "--kind", "rtp" if kind == "rtp8k" else "pcm", "--freq", str(freq), "--interval-ms", str(interval_ms), "--pattern", str(pat), "--n-pattern",
           str(n_pat), "--dump", str(dump), "--keep", str(keep), "--sample", ",".join(map(str, sample)), "--lat", str(lat), "--phases", str(phases), "--calls", str(calls), "--n-far", "16"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["ticks"] == ticks and d["group_ticks"] == ticks * phases and d["sub_batches"] == {1: 3, 2: 4, 3: 3, 4: 4}[phases] and d["failed_steps"] == 0 and d["far_end_per_stream"] == bool(calls) and d["rc"] == 0
    assert d["budget_ms"] == tick_ms - 2 and d["misses"] <= 3, d  # (a late wake-up or two of a shared test box are not the library's: DESIGN.md 5a)
    lat_ms = np.fromfile(lat, np.float64)
    assert lat_ms.size == ticks * phases and abs(np.percentile(lat_ms, 50) - d["p50_ms"]) < 1e-3
    got = np.fromfile(dump, rows.dtype).reshape(keep, len(sample), rows.shape[2])
    T = prime + ticks
    for col, s in enumerate(sample):
        want = bench.paced_replay(kind, far, rows, s % n_pat, T, interval_ms)[T - keep:]
        assert np.array_equal(got[:, col], want), (kind, s)


def test_bench_paced_line(cuda):
    """bench.py --paced: the same loop from Python; its line carries the distribution, the misses and an in-run parity proof"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--paced", "--streams", "4096", "--sub-batch", "1500", "--ticks", "100",
                        "--paced-prime", "80", "--tick-ms", "20"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    rt = d["realtime"]
    assert rt["streams"] == 4096 and rt["sub_batches"] == 3 and rt["ticks"] == 100 and rt["budget_ms"] == 18.0
    assert rt["parity_checked"]["max_lsb"] == 0 and rt["parity_checked"]["ticks_replayed"] == 180
    assert rt["misses"] <= 3 and rt["failed_steps"] == 0, rt
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--paced", "--resident", "--paced-kind", "rtp8k", "--streams", "2048", "--ticks", "60",
                        "--paced-prime", "60"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rt = json.loads(r.stdout.strip().splitlines()[-1])["realtime"]
    assert rt["parity_checked"]["max_lsb"] == 0 and rt["bytes_over_pcie_per_tick"] == 0 and rt["misses"] <= 3, rt
    # staggered release: four groups 5 ms apart, from host memory and resident; and with a far-end per stream
    for extra in ([], ["--resident"], ["--calls"], ["--calls", "--resident"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--paced", "--phases", "4", "--streams", "6000", "--sub-batch", "1000", "--ticks", "60",
                            "--paced-prime", "70"] + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        rt = json.loads(r.stdout.strip().splitlines()[-1])["realtime"]
        assert rt["phases"] == 4 and rt["ticks"] == 240 and rt["sub_batches"] == 8 and rt["parity_checked"]["max_lsb"] == 0 and rt["misses"] <= 3, rt
