#!/usr/bin/env python3
"""paced_host.py -- drive examples/host_paced (the paced heartbeat in C) on the GPU box and prove its rows.

    python tools_dev/paced_host.py --streams 393216,425984 --ticks 1500 [--phases 4] [--kind pcm16k|rtp8k] [--tick-ms 20] [--sub 32768]
                                   [--stop-at-miss] [--out profiles/r06/paced_x.jsonl]

Per stream count: writes the pattern file (bench.paced_pattern: `slots` ticks of 256 distinct streams), runs host_paced, replays 12
sampled streams through the oracle for every tick of the run (start-up included) and compares the rows host_paced kept for the last
ticks.  One JSON line per run (host_paced's own line + parity + the latency file's histogram)."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class DeviceWatch:
    """The device's shader clock, power and temperature every 2 s while host_paced runs (sysfs of this process's GPU; whatever cannot
    be read is left out): a long paced run is also a statement about what the power management does to a device at 80 - 98 % duty."""

    def __init__(self):
        import glob
        import threading
        import torch
        self.rows, self._stop = [], threading.Event()
        try:
            p = torch.cuda.get_device_properties(0)
            base = "/sys/bus/pci/devices/%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        except Exception:
            base = None
        self.sclk = base and os.path.join(base, "pp_dpm_sclk")
        hw = glob.glob(os.path.join(base, "hwmon", "hwmon*")) if base else []
        self.power = hw and os.path.join(hw[0], "power1_average")
        self.temp = hw and os.path.join(hw[0], "temp1_input")
        self.t0 = None
        self._thread = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _read(path):
        try:
            return open(path).read()
        except (OSError, TypeError):
            return None

    def _run(self):
        import time
        self.t0 = time.time()
        while not self._stop.wait(2.0):
            row = [round(time.time() - self.t0, 1), None, None, None]
            txt = self._read(self.sclk)
            if txt:
                for line in txt.splitlines():
                    if line.rstrip().endswith("*"):
                        try:
                            row[1] = int(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
                        except ValueError:
                            pass
            v = self._read(self.power)
            row[2] = round(int(v) / 1e6, 1) if v and v.strip().isdigit() else None
            v = self._read(self.temp)
            row[3] = round(int(v) / 1e3, 1) if v and v.strip().isdigit() else None
            self.rows.append(row)

    def start(self):
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        self._thread.join(timeout=5)
        out = {}
        for i, name in ((1, "sclk_mhz"), (2, "power_w"), (3, "temp_c")):
            v = [r[i] for r in self.rows if r[i] is not None]
            if v:
                out[name] = {"min": min(v), "median": float(np.median(v)), "max": max(v), "first_minute_median": float(np.median(v[:30])),
                             "last_minute_median": float(np.median(v[-30:]))}
        step = max(1, len(self.rows) // 60)
        out["series_t_sclk_power_temp"] = self.rows[::step]
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", required=True)
    ap.add_argument("--ticks", type=int, default=1500)
    ap.add_argument("--prime", type=int, default=150)
    ap.add_argument("--phases", type=int, default=1)
    ap.add_argument("--kind", default="pcm16k")
    ap.add_argument("--tick-ms", type=float, default=20.0)
    ap.add_argument("--interval-ms", type=int, default=0)
    ap.add_argument("--sub", type=int, default=32768)
    ap.add_argument("--slots", type=int, default=4)
    ap.add_argument("--keep", type=int, default=24)
    ap.add_argument("--stop-at-miss", action="store_true")
    ap.add_argument("--calls", type=int, default=0, help="1: every stream hears a far-end of its own (wmx_rt_create_pcm_calls), 16 distinct far signals")
    ap.add_argument("--rt-prio", type=int, default=0, help="host_paced asks for SCHED_FIFO at this priority (and mlockall); the line says whether it got it")
    ap.add_argument("--spin", type=int, default=0, help="1: host_paced never sleeps between ticks (spins on the clock)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import bench
    form, freq = bench.PACED_KINDS[a.kind]
    interval_ms = a.interval_ms or int(a.tick_ms)
    n_pat = 256
    far, rows = bench.paced_pattern(a.kind, a.slots, interval_ms, n_pattern=n_pat, n_far=16 if a.calls else 1)
    tmp = tempfile.mkdtemp(prefix="paced_")
    pat = os.path.join(tmp, "pattern.bin")
    with open(pat, "wb") as f:
        f.write(np.ascontiguousarray(far).tobytes())
        f.write(np.ascontiguousarray(rows).tobytes())
    host = os.path.join(ROOT, "examples", "host_paced")
    for S in [int(x) for x in a.streams.split(",")]:
        sample = sorted(set(int(i) for i in np.linspace(0, S - 1, 12)))
        dump, lat, lag = os.path.join(tmp, "dump.bin"), os.path.join(tmp, "lat.f64"), os.path.join(tmp, "lag.f64")
        cmd = [host, "--streams", str(S), "--sub", str(a.sub), "--slots", str(a.slots), "--tick-ms", str(a.tick_ms), "--ticks", str(a.ticks), "--prime",
               str(a.prime), "--kind", form, "--freq", str(freq), "--interval-ms", str(interval_ms), "--phases", str(a.phases), "--pattern", pat,
               "--n-pattern", str(n_pat), "--dump", dump, "--keep", str(a.keep), "--sample", ",".join(map(str, sample)), "--lat", lat, "--lag", lag, "--spin", str(a.spin), "--rt-prio", str(a.rt_prio), "--calls", str(a.calls), "--n-far", "16"]
        watch = DeviceWatch().start()
        r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)  # stderr passes through: a long run reports twice a minute
        device = watch.stop()
        if r.returncode != 0:
            sys.exit(1)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        got = np.fromfile(dump, rows.dtype).reshape(a.keep, len(sample), rows.shape[2])
        T = a.prime + a.ticks
        worst = 0
        for col, s in enumerate(sample):
            want = bench.paced_replay(a.kind, far, rows, s % n_pat, T, interval_ms)[T - a.keep:]
            worst = max(worst, int(np.abs(got[:, col].astype(np.int32) - want.astype(np.int32)).max()))
        lat_ms, lag_ms = np.fromfile(lat, np.float64), np.fromfile(lag, np.float64)
        # whose hiccup was it?  a slow group-tick (> median + 1.5 ms) started late (the host: its thread overslept, or a predecessor overran)
        # or ran long (the device / the link) -- or both
        svc = lat_ms - lag_ms
        slow = lat_ms > np.median(lat_ms) + 1.5
        d["slow_group_ticks"] = {"count": int(slow.sum()), "released_late_by_more_than_1ms": int((slow & (lag_ms > 1.0)).sum()),
                                 "ran_long_by_more_than_1ms": int((slow & (svc > np.median(svc) + 1.0)).sum()),
                                 "service_p50_ms": round(float(np.median(svc)), 4), "service_max_ms": round(float(svc.max()), 4),
                                 "release_lag_over_1ms": int((lag_ms > 1.0).sum())}
        if a.out:  # the series themselves beside the line (float32, group-tick order)
            np.stack([lat_ms, lag_ms]).astype(np.float32).tofile(a.out + ".S%d.lat_lag.f32" % S)
        edges = [0, 2, 4, 6, 8, 10, 12, 14, 15, 16, 17, 18, 19, 20, 25, 50, 1e9]
        d["latency_histogram_ms"] = {("%g-%g" % (edges[i], edges[i + 1])) if edges[i + 1] < 1e9 else (">%g" % edges[i]): int(c)
                                     for i, c in enumerate(np.histogram(lat_ms, edges)[0]) if c}
        d["parity_checked"] = {"streams": len(sample), "ticks_compared": a.keep, "ticks_replayed": T, "max_lsb": worst,
                               "oracle": "oracle/orc_*.c chain (port): every tick of the run replayed per sampled stream"}
        d["device"] = device
        d["spin_between_ticks"] = bool(a.spin)
        d["stream_frames_per_s_sustained"] = S * (interval_ms // 10) / (a.tick_ms * 1e-3)
        line = json.dumps(d)
        print(line)
        sys.stdout.flush()
        if a.out:
            with open(a.out, "a") as f:
                f.write(line + "\n")
        if a.stop_at_miss and d["misses"] > 0:
            break


if __name__ == "__main__":
    main()
