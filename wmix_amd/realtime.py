"""The paced heartbeat over S concurrent streams in host memory: a ctypes mirror of wmx_rt_* (wmix_amd/csrc/rt.hip) and the paced
loop around it.

The reference's record thread is a paced loop -- one package of WMIX_INTERVAL_MS = 20 ms per tick (src/wmixConf.h:112), the tick's work
and a sleep adding up to WMIX_INTERVAL_MS * 1000 - 2000 us (src/wmix.c:536-538, 820; the play thread :1468-1474): a tick has to be done
2 ms before the next package is due.  `RtBatch` is that tick for S streams (sub-batches of a wmx_pipe each, uploads and downloads beside
the compute); `paced_loop` releases one tick per period on an absolute schedule and reports, per tick, the time from the SCHEDULED
release to the moment the last row is back in host memory -- a tick that starts late because its predecessor overran carries that
backlog in its own latency.  examples/host_paced.c is the same loop in C.
"""
import ctypes as C
import time

import numpy as np
import torch

from ._lib import check, lib
from .chain import AEC, AGC, NS, VAD
from .pipeline import DATAGRAM, PKT, _host_rows


class RtBatch:
    """wmx_rt_create_pcm (kind "pcm": rows are int16 packages of chn x freq x interval_ms) or wmx_rt_create_rtp (kind "rtp": rows are
    172-byte RTP/PCMA datagrams, 8 kHz mono, 20 ms)."""

    def __init__(self, n_streams, dev, sub_batch=65536, slots=2, kind="pcm", chn=1, freq=16000, interval_ms=20, agc_value=5,
                 stages=NS | AEC | AGC | VAD, compute_streams=1, far_rows=False):
        """far_rows: every stream hears a far-end of its own (wmx_rt_create_pcm_calls: aec_process2's far-end is per handle) -- the far-end
        then comes in rows like the near-end: h_far_rows[b][slot] is [batch_n[b], package], a device far is [S, package]."""
        self.n, self.dev, self.kind, self.slots, self.far_rows = int(n_streams), dev, kind, slots, bool(far_rows)
        self._h = C.c_void_p()
        L = lib()
        if kind == "pcm":
            if far_rows:
                check(L.wmx_rt_create_pcm_calls(C.byref(self._h), self.n, sub_batch, slots, chn, freq, interval_ms, agc_value, stages), "wmx_rt_create_pcm_calls")
            else:
                check(L.wmx_rt_create_pcm(C.byref(self._h), self.n, sub_batch, slots, chn, freq, interval_ms, agc_value, stages), "wmx_rt_create_pcm")
            self.pkt10, self.ppc = freq // 100 * chn, interval_ms // 10
            self.row, self.row_dtype, self.far_shape = self.pkt10 * self.ppc, np.int16, (self.ppc, self.pkt10)
            self.row_bytes = self.row * 2
        else:
            check(L.wmx_rt_create_rtp(C.byref(self._h), self.n, sub_batch, slots, 0, agc_value, stages), "wmx_rt_create_rtp")
            self.pkt10, self.ppc = PKT, 2
            self.row, self.row_dtype, self.far_shape, self.row_bytes = DATAGRAM, np.uint8, (2, PKT), DATAGRAM
        check(L.wmx_rt_set_compute_streams(self._h, compute_streams), "wmx_rt_set_compute_streams")
        self.B = L.wmx_rt_batches(self._h)
        self.batch_n = [L.wmx_rt_batch_streams(self._h, b) for b in range(self.B)]
        self.lo = np.concatenate([[0], np.cumsum(self.batch_n)]).astype(np.int64)
        self.pipes = [L.wmx_rt_pipe(self._h, b) for b in range(self.B)]
        # numpy views of the library's pinned rows: h_in[b][slot] is [batch_n[b], row]
        self.h_in = [[_host_rows(L.wmx_pipe_in(p, s), (n, self.row), self.row_dtype) for s in range(slots)] for p, n in zip(self.pipes, self.batch_n)]
        self.h_out = [[_host_rows(L.wmx_pipe_out(p, s), (n, self.row), self.row_dtype) for s in range(slots)] for p, n in zip(self.pipes, self.batch_n)]
        self.h_far = [_host_rows(L.wmx_rt_far(self._h, s), self.far_shape, np.int16) for s in range(slots)] if not far_rows else None
        self.h_far_rows = ([[_host_rows(L.wmx_pipe_far(p, s), (n, self.row), np.int16) for s in range(slots)] for p, n in zip(self.pipes, self.batch_n)]
                           if far_rows else None)

    def locate(self, stream):
        """(sub-batch, row) of a stream"""
        b = int(np.searchsorted(self.lo, stream, side="right") - 1)
        return b, int(stream - self.lo[b])

    def fill(self, slot, rows):
        """rows: [S, row] host array -> the slot's pinned input rows of every sub-batch"""
        for b in range(self.B):
            self.h_in[b][slot][:] = rows[self.lo[b]:self.lo[b + 1]]

    def fill_far(self, slot, rows):
        """far_rows: [S, package] host array -> the slot's pinned far rows of every sub-batch"""
        for b in range(self.B):
            self.h_far_rows[b][slot][:] = rows[self.lo[b]:self.lo[b + 1]]

    def gather(self, slot, streams=None):
        if streams is None:
            return np.concatenate([self.h_out[b][slot] for b in range(self.B)])
        return np.stack([self.h_out[b][slot][r] for b, r in map(self.locate, streams)])

    def _far(self, far):
        if far is None:
            return None
        assert far.is_cuda and far.dtype == torch.int16 and far.is_contiguous()
        assert far.numel() == self.ppc * self.pkt10 * (self.n if self.far_rows else 1)
        return far.data_ptr()

    def submit(self, far=None):
        slot = C.c_int(-1)
        check(lib().wmx_rt_submit(self._h, self._far(far), C.byref(slot), torch.cuda.current_stream().cuda_stream), "wmx_rt_submit")
        return slot.value

    def wait(self):
        check(lib().wmx_rt_wait(self._h), "wmx_rt_wait")

    def poll(self):
        """non-blocking wait(): True when every row of every queued tick is in host memory"""
        rc = lib().wmx_rt_poll(self._h)
        if rc < 0:
            check(rc, "wmx_rt_poll")
        return rc == 1

    def tick(self, far=None):
        """one tick: every sub-batch up, through the chain and down again; returns (the slot) when the last row is in host memory"""
        slot = C.c_int(-1)
        check(lib().wmx_rt_tick(self._h, self._far(far), C.byref(slot), torch.cuda.current_stream().cuda_stream), "wmx_rt_tick")
        return slot.value

    def step_resident(self, rows, far, out=None):
        """the tick's launches alone, rows [S, row] on the device (PCM: in place; RTP: datagrams into `out`)"""
        assert rows.is_cuda and rows.stride(1) == 1 and rows.shape[0] == self.n
        es = rows.element_size()
        o = rows if out is None else out
        check(lib().wmx_rt_step_resident(self._h, rows.data_ptr(), rows.stride(0) * es, self._far(far), o.data_ptr(), o.stride(0) * o.element_size(),
                                         torch.cuda.current_stream().cuda_stream), "wmx_rt_step_resident")

    def failed_steps(self):
        return sum(lib().wmx_pipe_failed_steps(p) for p in self.pipes)

    def close(self):
        if self._h:
            lib().wmx_rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GpuClock:
    """The shader clock the driver reports for THIS process's device, from sysfs (pp_dpm_sclk marks the current level with '*'; an idle
    device shows the sleep level "S: 94Mhz"); None where it cannot be read.  The box's sysfs lists every GPU of the host: the device is
    found by its PCI address.  Read off the critical path: just before a tick is released."""

    def __init__(self, index=0):
        self.path = None
        try:
            p = torch.cuda.get_device_properties(index)
            bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
            cand = ["/sys/bus/pci/devices/%s/pp_dpm_sclk" % bdf]
        except Exception:
            cand = []
        for c in cand:
            try:
                open(c).read()
                self.path = c
                break
            except OSError:
                continue

    def mhz(self):
        if not self.path:
            return None
        try:
            for line in open(self.path).read().splitlines():
                if line.rstrip().endswith("*"):
                    return int(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
        except (OSError, ValueError, IndexError):
            return None
        return None

    # The read goes through the driver to the SMU and can take milliseconds: never on the releasing thread.  start() samples every
    # `period_s` on a thread of its own (file I/O releases the GIL); stop() returns the samples.
    def start(self, period_s=0.25):
        import threading
        self._samples, self._stop = [], threading.Event()

        def run():
            while not self._stop.wait(period_s):
                v = self.mhz()
                if v is not None:
                    self._samples.append(v)
        self._thread = threading.Thread(target=run, daemon=True)
        if self.path:
            self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread.is_alive():
            self._thread.join(timeout=2.0)
        return list(self._samples)


def paced_loop(tick_fn, tick_ms, n_ticks, clock=None, clock_every=16, spin_us=300, after=None):
    """Release tick k at t0 + k * tick_ms (absolute schedule): sleep until shortly before, spin the rest, call tick_fn(k), which returns
    when the tick's last row is in host memory.  Returns (latency_ms[k] = completion - scheduled release, lag_ms[k] = actual start -
    scheduled release, sclk MHz samples taken every 250 ms by a thread of their own)."""
    period = tick_ms * 1e-3
    lat, lag = np.empty(n_ticks), np.empty(n_ticks)
    now = time.perf_counter
    if clock is not None:
        clock.start()
    t0 = now() + period
    for k in range(n_ticks):
        due = t0 + k * period
        d = due - now() - spin_us * 1e-6
        if d > 0:
            time.sleep(d)
        while now() < due:
            pass
        start = now()
        tick_fn(k)
        end = now()
        lag[k], lat[k] = (start - due) * 1e3, (end - due) * 1e3
        if after is not None:
            after(k)  # behind the clock: whatever the host does with the rows is not part of the tick
    return lat, lag, (clock.stop() if clock is not None else [])


def paced_groups(submit, poll, wait, n_groups, tick_ms, n_ticks, clock=None, after=None):
    """Staggered release: group g (of n_groups groups of streams, each with the whole tick as its period) is released at
    t0 + (k * n_groups + g) * tick_ms / n_groups -- the streams of a server do not all deliver their package at the same instant, and a
    device that works in n_groups short bursts per period never idles long enough for its power management to clock it down
    (profiles/r06/README_paced.md).  submit(g) queues the group's tick and returns; poll(g) is the non-blocking completion check;
    wait(g) blocks.  Between releases the loop polls the groups in flight, so a completion is seen within microseconds of the last
    row's arrival.  Returns (latency_ms, lag_ms, clock samples) over all n_ticks * n_groups group-ticks, in release order."""
    P, sub = n_groups, tick_ms * 1e-3 / n_groups
    total = n_ticks * P
    lat, lag = np.empty(total), np.empty(total)
    now = time.perf_counter
    flying = {}  # group -> (index, due)
    if clock is not None:
        clock.start()

    def reap(block_group=None):
        for g in list(flying):
            if g == block_group:
                wait(g)
            elif not poll(g):
                continue
            j, due = flying.pop(g)
            lat[j] = (now() - due) * 1e3
            if after is not None:
                after(j, g)

    t0 = now() + tick_ms * 1e-3
    for j in range(total):
        due, g = t0 + j * sub, j % P
        while now() < due:
            if flying:
                reap()
            elif due - now() > 4e-4:
                time.sleep(due - now() - 3e-4)
        if g in flying:  # its previous tick is not back yet: the release waits for it (and the wait counts)
            reap(block_group=g)
        start = now()
        submit(g)
        flying[g] = (j, due)
        lag[j] = (start - due) * 1e3
    while flying:
        reap(block_group=next(iter(flying)))
    return lat, lag, (clock.stop() if clock is not None else [])


def latency_summary(lat_ms, lag_ms, tick_ms, clk=None):
    """p50 / p99 / p99.9 / max and the misses against the reference's own budget: tick_ms - 2 ms (src/wmix.c:538)."""
    budget = tick_ms - 2.0
    q = np.percentile(lat_ms, [50, 99, 99.9])
    out = {"ticks": int(lat_ms.size), "tick_ms": tick_ms, "budget_ms": budget, "p50_ms": round(float(q[0]), 4), "p99_ms": round(float(q[1]), 4),
           "p99_9_ms": round(float(q[2]), 4), "max_ms": round(float(lat_ms.max()), 4), "misses": int((lat_ms > budget).sum()),
           "overruns_of_the_period": int((lat_ms > tick_ms).sum()),
           "release_lag_p50_ms": round(float(np.percentile(lag_ms, 50)), 4), "release_lag_max_ms": round(float(lag_ms.max()), 4),
           # the tick's own duration (actual start -> done), whatever backlog it started with
           "service_p50_ms": round(float(np.percentile(lat_ms - lag_ms, 50)), 4), "service_max_ms": round(float((lat_ms - lag_ms).max()), 4),
           "worst_tick": int(lat_ms.argmax())}
    if clk:
        out["sclk_mhz_sampled"] = {"min": int(min(clk)), "median": int(np.median(clk)), "max": int(max(clk)), "samples": len(clk)}
    return out
