#!/usr/bin/env python3
"""bench.py -- throughput of the wmix DSP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

--gpus N > 1 outside a torchrun environment starts N rank processes of this script itself (one per GPU, RCCL);
under torchrun (WORLD_SIZE set) --gpus must equal WORLD_SIZE.

A "step" is one pass of the hot path over one batch of synthetic 10 ms frames that
is already resident in HBM.  Streams shard across ranks with no data-path
collective except the one RCCL broadcast of the shared AEC far-end frame
(SURVEY.md section 8e), so scaling is weak: every rank owns the full per-GPU batch.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying
`roofline` (dominant kernel, HIP-event timed) and `cpu_baseline` (the oracle
timed on this box's host cores, rank 0 / N=1 only, bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


# ----------------------------------------------------------------------------- workloads
class G711Workload:
    """configs[0]-style plumbing case scaled up: mu-law encode + decode round trip of
    80-sample (10 ms @ 8 kHz) frames.  Algorithmic bytes per frame = 160 + 80 (encode)
    + 80 + 160 (decode) = 480 B."""
    name = "g711_ulaw_roundtrip_8k"
    pmc_tag = "g711"
    dtype = "u8"
    frame_samples = 80
    bytes_per_frame = 480.0
    dominant_kernel = "g711_encode_kernel<1>"
    dominant_bytes_per_frame = 240.0  # encode: 160 B in + 80 B out

    def __init__(self, dev, n_streams, rank):
        from wmix_amd import g711, synth
        self.g711 = g711
        self.n_frames = n_streams  # one 10 ms frame per stream per step
        pcm = synth.lcg_noise(1000 + rank * 7919 + np.arange(64, dtype=np.uint64), self.n_frames * 80 // 64, 20000)
        self.pcm = torch.from_numpy(pcm.reshape(-1)).to(dev)
        self.code = torch.empty(self.pcm.numel(), dtype=torch.uint8, device=dev)
        self.back = torch.empty_like(self.pcm)
        self.ev = []

    def step(self, timed):
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self.g711.encode("u", self.pcm, self.code)
        if timed:
            e1.record()
            self.ev.append((e0, e1))
        self.g711.decode("u", self.code, self.back)

    def dominant_ms(self):
        return float(np.mean([a.elapsed_time(b) for a, b in self.ev])) if self.ev else None

    def config(self):
        return {"workload": self.name, "frames_per_step_per_gpu": self.n_frames, "frame": "80 x int16 (10 ms @ 8 kHz mono)"}

    def parity_check(self):
        """EVERY code and EVERY decoded sample of the last step against oracle/orc_g711.c (stateless: each step writes the same
        buffers from the same input), both laws of the round trip timed; bit-exact is the bar."""
        from oracle import loader
        port = loader.port()
        pcm = np.ascontiguousarray(self.pcm.cpu().numpy())
        want_code, want_back = np.zeros(pcm.size, np.uint8), np.zeros(pcm.size, np.int16)
        CH = 1 << 26  # samples per oracle call: the reference's DataLen is an int (bytes), and a step may hold 2^30 samples
        for a in range(0, pcm.size, CH):
            n = min(CH, pcm.size - a)
            assert port.orc_PCM2G711u(C.c_void_p(pcm[a:].ctypes.data), C.c_void_p(want_code[a:].ctypes.data), n * 2, 0) == n
            assert port.orc_G711u2PCM(C.c_void_p(want_code[a:].ctypes.data), C.c_void_p(want_back[a:].ctypes.data), n, 0) == n * 2
        bad_c = int((self.code.cpu().numpy() != want_code).sum())
        bad_p = int((self.back.cpu().numpy() != want_back).sum())
        return {"frames": self.n_frames, "samples_compared": 2 * int(pcm.size), "codes_differing": bad_c, "decoded_differing": bad_p,
                "max_lsb": 0 if bad_c + bad_p == 0 else 1 << 15, "oracle": "oracle/orc_g711.c (port), every sample of the step"}

    def cpu_baseline(self, budget_s):
        from oracle import loader
        port = loader.port()
        n = 80 * 20000
        pcm = np.ascontiguousarray(self.pcm[:n].cpu().numpy())
        code = np.zeros(n, np.uint8)
        back = np.zeros(n, np.int16)
        enc, dec = port.orc_PCM2G711u, port.orc_G711u2PCM

        def one():
            c, bk = np.zeros(n, np.uint8), np.zeros(n, np.int16)
            enc(C.c_void_p(pcm.ctypes.data), C.c_void_p(c.ctypes.data), n * 2, 0)
            dec(C.c_void_p(c.ctypes.data), C.c_void_p(bk.ctypes.data), n, 0)
        reps, v1, nc, vn = _cpu_rates(one, 20000, budget_s)
        return {"value": v1, "unit": "frames/s", "cores": 1, "kind": "port", "all_cores_value": vn, "all_cores": nc, "cpu_model": _cpu_model(),
                "sample": "%d x 20000 frames of 80 samples through oracle/orc_g711.c (-O2), 1 thread; then all %d cores" % (reps, nc)}


class _StageTimer:
    """HIP-event timing of individual launches on torch's current stream (the stream every wmx_* call is given).

    An event pair around a launch is not free (about 4 us of GPU idle per pair on MI355X, tools_dev/event_overhead.py:
    eight pairs cost the chain's step 2 %), so inside the timed region (timed=True) only the DOMINANT stage -- the one
    the roofline entry is computed from -- is bracketed; the per-stage breakdown comes from a few extra steps after
    the timed region (timed="all")."""

    def __init__(self, dominant):
        self.dominant = dominant
        self.ev = {}      # timed region: dominant stage only
        self.ev_all = {}  # breakdown pass: every stage

    def run(self, name, timed, fn):
        if not timed or (timed is True and name != self.dominant):
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        (self.ev_all if timed == "all" else self.ev).setdefault(name, []).append((e0, e1))
        return r

    @staticmethod
    def _mean(v):
        return float(np.mean([a.elapsed_time(b) for a, b in v])) if v else None

    def dominant_ms(self):
        return self._mean(self.ev.get(self.dominant))

    def mean_ms(self, name):
        """Per-stage figure of the breakdown pass: the mean of its (up to 16) launches without outliers beyond five times
        their median -- bracketing every launch with events lets a host-side hiccup between two records show up as one
        40 ms "launch" now and then (seen on the ns_agc_mix_32k line: one among 16, on a different stage each run; the timed
        region has no such gaps).  A plain median would misreport the AEC, whose launches alternate between two block counts."""
        v = self.ev_all.get(name)
        if v:
            t = np.array([a.elapsed_time(b) for a, b in v])
            return float(t[t <= 5 * np.median(t)].mean())
        return self._mean(self.ev.get(name))


class NsWorkload:
    """BASELINE.json configs[1]: WebRtcNs_Process, 16 kHz mono, 4096 streams per GPU, one 10 ms packet per stream
    per step.  Algorithmic bytes per stream-frame = 320 in + 320 out + 2 x 12 200 live state = 25 040 B."""
    name = "ns_16k_mono"
    pmc_tag = "ns"
    dtype = "f32"
    bytes_per_frame = 25040.0
    dominant_kernel = "ns_kernel<256, 1>"
    dominant_bytes_per_frame = 25040.0
    freq, pkt = 16000, 160

    def __init__(self, dev, n_streams, rank):
        from wmix_amd import synth
        from wmix_amd.ns import NsBatch
        self.n_frames = n_streams
        # one full gate period of the SURVEY 8d recipe (noise + a tone gated every 100 frames), 256 distinct streams, tiled
        self.K = 200
        base = synth.ns_input(2000 + 7919 * rank, 256, self.K, self.pkt).reshape(256, self.K, self.pkt)
        self.base = base
        b = torch.from_numpy(np.ascontiguousarray(base.transpose(1, 0, 2))).to(dev)  # [K, 256, pkt]
        self.inp = b[:, torch.arange(n_streams, device=dev) % 256]                     # [K, S, pkt] packet-major
        self.work = torch.empty_like(self.inp[0:1])
        self.ns = NsBatch(n_streams, 1, self.freq)
        self.t = _StageTimer("ns")
        self.k = 0
        self.sample = [int(i) for i in np.linspace(0, n_streams - 1, 16)]
        self.rec = []

    def step(self, timed):
        k = self.k % self.K
        self.t.run("ns", timed, lambda: self.ns.process_packet_major(self.inp[k:k + 1], self.work))
        if timed is not True:  # outside the timed region: keep what the sampled streams produced, for parity_check()
            self.rec.append((self.k, self.work[0, self.sample].clone()))
        self.k += 1

    def parity_check(self):
        """Replays 16 sampled streams through the oracle's NS for exactly the packets this run fed and compares every packet
        recorded outside the timed region (those behind it depend on every timed step through the noise model).  Float path:
        bit-exact."""
        from oracle import loader
        port = loader.port()
        worst, n, n_off = 0, 0, 0
        for col, s in enumerate(self.sample):
            x = np.concatenate([self.base[s % 256, k % self.K] for k in range(self.k)])
            want = loader.run_ns(port, 1, self.freq, x, self.pkt, prefix="orc").reshape(self.k, self.pkt)
            for k, got in self.rec:
                d = np.abs(got[col].cpu().numpy().astype(np.int32) - want[k].astype(np.int32))
                worst, n, n_off = max(worst, int(d.max())), n + 1, n_off + int((d > 0).sum())
        return {"streams": len(self.sample), "packets_compared": n, "max_lsb": worst, "samples_off_by_one": n_off,
                "oracle": "oracle/orc_ns.c (port)", "steps_replayed": self.k}

    def dominant_ms(self):
        return self.t.dominant_ms()

    def stage_ms(self):
        return {"ns": self.t.mean_ms("ns")}

    def config(self):
        return {"workload": self.name, "streams_per_gpu": self.n_frames, "frame": "160 x int16 (10 ms @ 16 kHz mono)",
                "sum_order": "reference (the only one the library has)",
                "input": "noise A=3000 + 3000 sin(0.01 t) gated every 100 frames (SURVEY 8d recipe), 256 distinct streams x %d "
                         "packets, tiled" % self.K}

    def cpu_baseline(self, budget_s):
        from oracle import loader
        port = loader.port()
        n = 3000
        x = np.ascontiguousarray(self.base[0].reshape(-1))
        x = np.tile(x, n // self.K + 1)[: n * self.pkt]
        reps, v1, nc, vn = _cpu_rates(lambda: loader.run_ns(port, 1, self.freq, x, self.pkt, prefix="orc"), n, budget_s)
        return {"value": v1, "unit": "frames/s", "cores": 1, "kind": "port", "all_cores_value": vn, "all_cores": nc, "cpu_model": _cpu_model(),
                "sample": "%d x %d packets of one 16 kHz stream through oracle/orc_ns.c (-O2), 1 thread; then all %d cores" % (reps, n, nc)}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _usable_cores():
    """Host cores this process may actually use: the scheduler affinity mask, cut down to the cgroup CPU quota (a GPU box
    hands a 1-GPU job a share of the host, not all of its cores)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _cpu_rates(one_call, frames_per_call, budget_s):
    """Time `one_call()` (one self-contained run of the CPU path over `frames_per_call` stream-frames; a ctypes call, so the
    GIL is released while it runs) on ONE thread, then on every host core at once (one independent stream per thread --
    streams never interact).  Returns (reps_1, frames/s on 1 core, n_cores, frames/s on all cores)."""
    import threading
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < budget_s * 0.4 or reps == 0:
        one_call()
        reps += 1
    v1 = reps * frames_per_call / (time.perf_counter() - t0)
    n = _usable_cores()
    counts = [0] * n
    stop_at = time.perf_counter() + budget_s * 0.6

    def worker(i):
        while time.perf_counter() < stop_at or counts[i] == 0:
            one_call()
            counts[i] += 1

    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    vn = sum(counts) * frames_per_call / (time.perf_counter() - t0)
    return reps, v1, n, vn


class NsxWorkload:
    """the fixed-point noise suppressor (the reference's MAKE_WEBRTC_NSX build of ns_process), 16 kHz mono,
    65 536 streams per GPU, one 10 ms packet per stream per step.  Algorithmic bytes per stream-frame = 320 in + 320 out +
    2 x 5 516 live state (NoiseSuppressionFixedC's per-frame fields: two 256-sample int16 buffers, seven int16 and four
    int32 per-bin arrays of 129, ~100 B of scalars; nsx_core.h:23-123) = 11 672 B."""
    name = "nsx_16k_mono"
    dtype = "int16/int32 (fixed point)"
    bytes_per_frame = 11672.0
    dominant_kernel = "nsx_kernel<256, 1>"
    dominant_bytes_per_frame = 11672.0
    pmc_tag = "nsx"
    freq, pkt = 16000, 160

    def __init__(self, dev, n_streams, rank):
        from wmix_amd import synth
        from wmix_amd.nsx import NsxBatch
        self.n_frames = n_streams
        self.K = 200
        base = synth.ns_input(2100 + 7919 * rank, 256, self.K, self.pkt).reshape(256, self.K, self.pkt)
        self.base = base
        b = torch.from_numpy(np.ascontiguousarray(base.transpose(1, 0, 2))).to(dev)  # [K, 256, pkt]
        self.inp = b[:, torch.arange(n_streams, device=dev) % 256]                     # [K, S, pkt] packet-major
        self.work = torch.empty_like(self.inp[0:1])
        self.nsx = NsxBatch(n_streams, 1, self.freq)
        self.t = _StageTimer("nsx")
        self.k = 0
        self.sample = [int(i) for i in np.linspace(0, n_streams - 1, 16)]
        self.rec = []

    def step(self, timed):
        k = self.k % self.K
        self.t.run("nsx", timed, lambda: self.nsx.process_packet_major(self.inp[k:k + 1], self.work))
        if timed is not True:  # outside the timed region: keep what the sampled streams produced, for parity_check()
            self.rec.append((self.k, self.work[0, self.sample].clone()))
        self.k += 1

    def dominant_ms(self):
        return self.t.dominant_ms()

    def stage_ms(self):
        return {"nsx": self.t.mean_ms("nsx")}

    def config(self):
        return {"workload": self.name, "streams_per_gpu": self.n_frames, "frame": "160 x int16 (10 ms @ 16 kHz mono)",
                "input": "noise A=3000 + 3000 sin(0.01 t) gated every 100 frames (SURVEY 8d recipe), 256 distinct streams x %d "
                         "packets, tiled" % self.K}

    def parity_check(self):
        """Replays the sampled streams through the oracle for exactly the packets this run fed and compares every packet
        recorded outside the timed region (the ones behind it depend on every timed step through the state)."""
        from oracle import loader
        port = loader.port()
        worst, n = 0, 0
        for col, s in enumerate(self.sample):
            x = np.concatenate([self.base[s % 256, k % self.K] for k in range(self.k)])
            want = loader.run_nsx(port, 1, self.freq, x, self.pkt, prefix="orc").reshape(self.k, self.pkt)
            for k, got in self.rec:
                d = np.abs(got[col].cpu().numpy().astype(np.int32) - want[k].astype(np.int32)).max()
                worst, n = max(worst, int(d)), n + 1
        return {"streams": len(self.sample), "packets_compared": n, "max_lsb": worst, "oracle": "oracle/orc_nsx.c (port)",
                "steps_replayed": self.k}

    def cpu_baseline(self, budget_s):
        from oracle import loader
        n = 2000
        x = np.tile(np.ascontiguousarray(self.base[0].reshape(-1)), n // self.K + 1)[: n * self.pkt]
        lib, kind, prefix = loader.port(), "port", "orc"
        if loader.have_ref():
            try:
                lib, kind, prefix = loader.ref(), "reference", "ref"
                getattr(lib, "ref_run_nsx")
            except Exception:
                lib, kind, prefix = loader.port(), "port", "orc"
        reps, v1, nc, vn = _cpu_rates(lambda: loader.run_nsx(lib, 1, self.freq, x, self.pkt, prefix=prefix), n, budget_s)
        return {"value": v1, "unit": "frames/s", "cores": 1, "kind": kind, "all_cores_value": vn, "all_cores": nc,
                "cpu_model": _cpu_model(),
                "sample": "%d x %d packets of one 16 kHz stream through %s, 1 thread; then one stream per thread on all cores"
                          % (reps, n, "ns_process built with MAKE_WEBRTC_NSX (oracle/_ref/libwmixref.so, -O2)" if kind == "reference"
                             else "oracle/orc_nsx.c (-O2)")}


class AecmWorkload:
    """the fixed-point echo canceller (the reference's AECM build of aec_process2), 16 kHz mono, 65 536 near-end
    streams per GPU against one shared far-end, one 10 ms packet (two 80-sample frames, 2-3 blocks) per stream per step.
    Algorithmic bytes per stream-frame = 320 in + 320 out + 2 x 3 900 live per-stream state (AecmCore without the far-end
    history, which is shared: frame rings, dBufNoisy, outBuf, the three channel arrays, echoFilt / nearFilt / noiseEst and
    its counters, the three 64-entry log-energy histories, near mean spectrum + mean_bit_counts of the delay estimator;
    aecm_core.h:31-133, delay_estimator.h:24-73) = 8 440 B."""
    name = "aecm_16k_mono"
    dtype = "int16/int32 (fixed point)"
    bytes_per_frame = 8440.0
    dominant_kernel = "aecm_near_kernel"
    dominant_bytes_per_frame = 8440.0
    pmc_tag = "aecm"
    freq, pkt = 16000, 160

    def __init__(self, dev, n_streams, rank, dist=None, packets=1):
        from wmix_amd import synth
        from wmix_amd.aecm import AecmBatch
        from wmix_amd.shard import broadcast_far
        self._bcast = broadcast_far
        self.n_frames = n_streams
        self.dist, self.rank = dist, rank
        self.K = 200
        far = synth.far_end(3000, self.K, self.pkt)
        base = synth.near_end(3001 + 7919 * rank, 256, self.K, self.pkt, far=far).reshape(256, self.K, self.pkt)
        self.base, self.far_host = base, far.reshape(self.K, self.pkt)
        b = torch.from_numpy(np.ascontiguousarray(base.transpose(1, 0, 2))).to(dev)
        self.inp = b[:, torch.arange(n_streams, device=dev) % 256]
        self.far_src = torch.from_numpy(self.far_host.copy()).to(dev)
        self.far = torch.zeros(1, self.pkt, dtype=torch.int16, device=dev)
        self.work = torch.empty_like(self.inp[0:1])
        self.aecm = AecmBatch(n_streams, 1, self.freq, 10)
        self.t = _StageTimer("aecm")
        self.k = 0
        self.sample = [int(i) for i in np.linspace(0, n_streams - 1, 16)]
        self.rec = []

    def step(self, timed):
        k = self.k % self.K
        far = self.far_src[k:k + 1]
        if self.dist is not None:
            if self.rank == 0:
                self.far.copy_(far)
            self._bcast(self.far, self.dist, src=0)
            far = self.far
        self.work.copy_(self.inp[k:k + 1])
        self.t.run("aecm", timed, lambda: self.aecm.process2_packet_major(far, self.work))
        if timed is not True:
            self.rec.append((self.k, self.work[0, self.sample].clone()))
        self.k += 1

    def dominant_ms(self):
        return self.t.dominant_ms()

    def stage_ms(self):
        return {"aecm (far + near kernels)": self.t.mean_ms("aecm")}

    def config(self):
        return {"workload": self.name, "streams_per_gpu": self.n_frames, "frame": "160 x int16 (10 ms @ 16 kHz mono)",
                "far_end": "shared" + (", RCCL broadcast from rank 0 each step" if self.dist is not None else ", resident in HBM"),
                "input": "SURVEY 8d recipe (far noise A=8000, near = far delayed 40 / 2 + noise + gated tone), 256 distinct streams x "
                         "%d packets, tiled; the timed launch includes a copy of the input packet (in-place kernel)" % self.K}

    def parity_check(self):
        from oracle import loader
        port = loader.port()
        far = np.concatenate([self.far_host[k % self.K] for k in range(self.k)])
        worst, n = 0, 0
        for col, s in enumerate(self.sample):
            near = np.concatenate([self.base[s % 256, k % self.K] for k in range(self.k)])
            want = loader.run_aecm(port, 1, self.freq, 10, far, near, self.pkt, prefix="orc").reshape(self.k, self.pkt)
            for k, got in self.rec:
                d = np.abs(got[col].cpu().numpy().astype(np.int32) - want[k].astype(np.int32)).max()
                worst, n = max(worst, int(d)), n + 1
        return {"streams": len(self.sample), "packets_compared": n, "max_lsb": worst, "oracle": "oracle/orc_aecm.c (port)",
                "steps_replayed": self.k}

    def cpu_baseline(self, budget_s):
        from oracle import loader
        n = 2000
        far = np.tile(self.far_host.reshape(-1), n // self.K + 1)[: n * self.pkt]
        near = np.tile(np.ascontiguousarray(self.base[0].reshape(-1)), n // self.K + 1)[: n * self.pkt]
        lib, kind, prefix = loader.port(), "port", "orc"
        if loader.have_ref():
            try:
                lib, kind, prefix = loader.ref(), "reference", "ref"
                getattr(lib, "ref_run_aecm")
            except Exception:
                lib, kind, prefix = loader.port(), "port", "orc"
        reps, v1, nc, vn = _cpu_rates(lambda: loader.run_aecm(lib, 1, self.freq, 10, far, near, self.pkt, prefix=prefix), n, budget_s)
        return {"value": v1, "unit": "frames/s", "cores": 1, "kind": kind, "all_cores_value": vn, "all_cores": nc, "cpu_model": _cpu_model(),
                "sample": "%d x %d packets of one 16 kHz stream through %s, 1 thread; then one stream per thread on all %d cores"
                          % (reps, n, "aec_process2 built with the AECM switch (oracle/_ref/libwmixref.so, -O2)" if kind == "reference"
                             else "oracle/orc_aecm.c (-O2)", nc)}


class RtpChainWorkload:
    """SURVEY 8f-1, the packet edge end to end: per stream and step one 172-byte RTP/PCMA datagram (20 ms at 8 kHz) ->
    decode -> NS -> AEC -> AGC -> VAD (two 10 ms packets each) -> encode -> one 172-byte datagram (wmix_amd/pipeline.py,
    src/wmixTask.c:1278-1316 / 1124-1143 around src/wmix.c:613-709).  `value` is the resident rate (datagrams in and out in
    HBM); `pcie_inclusive` on the line is the same pipeline with the datagrams starting and ending in pinned host memory,
    copies on their own HIP streams overlapped with compute (three slots in flight).
    Algorithmic bytes per 10 ms stream-frame = (172 in + 172 out + 2 x (6 000 NS + 11 700 AEC + 668 AGC + 736 VAD) state) / 2
    = 19 276 B; dominant kernel = the AEC near kernel over two packets: 320 + 320 + 2 x 11 700 = 24 040 B per stream."""
    name = "rtp_chain_8k_pcma"
    dtype = "u8 datagrams, f32 / int16 chain"
    bytes_per_frame = 19276.0
    dominant_kernel = "aec_near_kernel<1>"
    dominant_bytes_per_frame = 12020.0  # per 10 ms frame: the launch covers two
    pmc_tag = "rtp_chain"

    def __init__(self, dev, n_streams, rank):
        from wmix_amd import g711, synth
        from wmix_amd.pipeline import RtpChain, StreamingPipe
        self.S, self.n_frames = n_streams, 2 * n_streams
        self.K = 50
        far = synth.far_end(5000, 2 * self.K, 80)
        base = synth.near_end(5001 + 7919 * rank, 256, 2 * self.K, 80, far=far).reshape(256, self.K, 160)
        self.far_host = far
        self.far = torch.from_numpy(far.reshape(self.K, 2, 80).copy()).to(dev)
        pcm = torch.from_numpy(np.ascontiguousarray(base.transpose(1, 0, 2))).to(dev)  # [K, 256, 160]
        codes = torch.empty(pcm.shape, dtype=torch.uint8, device=dev)
        g711.encode("a", pcm.reshape(-1), codes.reshape(-1))
        dg = torch.zeros((self.K, 256, 172), dtype=torch.uint8, device=dev)
        dg[:, :, 0], dg[:, :, 1] = 0x80, 0x88
        dg[:, :, 12:] = codes
        self.base_dg = dg.cpu().numpy()  # [K, 256, 172] for the oracle replay
        self.d_in = dg[:, torch.arange(n_streams, device=dev) % 256].contiguous()  # [K, S, 172]
        self.d_out = torch.zeros((n_streams, 172), dtype=torch.uint8, device=dev)
        self.chain = RtpChain(n_streams, dev)
        self.pipe_cls = StreamingPipe
        self.k = 0
        self.t = _StageTimer("step")
        self.sample = [int(i) for i in np.linspace(0, n_streams - 1, 8)]
        self.rec = []

    def step(self, timed):
        k = self.k % self.K
        self.t.run("step", timed, lambda: self.chain.step(self.d_in[k], self.far[k], self.d_out))
        if timed is not True:
            self.rec.append((self.k, self.d_out[self.sample].clone()))
        self.k += 1

    def dominant_ms(self):
        return None  # the line's roofline entry is filled from the whole step below (several kernels share it)

    def stage_ms(self):
        return {"ingest + ns + aec + agc + vad + egress": self.t.mean_ms("step")}

    def config(self):
        return {"workload": self.name, "streams_per_gpu": self.S, "frame": "172-byte RTP/PCMA datagram = 160 x int16 (20 ms @ 8 kHz mono)",
                "frames_per_step_per_gpu": self.n_frames, "pcie_inclusive": getattr(self, "pcie", None)}

    def measure_pcie(self, steps):
        """The same steps with host-resident datagrams: H2D of step k+1 and D2H of step k-1 overlap the compute of step k."""
        pipe = self.pipe_cls(self.chain)
        host = self.d_in[: pipe.SLOTS].cpu().numpy()
        for s in range(pipe.SLOTS):
            pipe.h_in[s][:] = host[s % self.K]
        for _ in range(4):
            pipe.submit(self.far[self.k % self.K])
            self.k += 1
        pipe.drain()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.submit(self.far[self.k % self.K])
            self.k += 1
        pipe.drain()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        self.pcie = {"value": self.n_frames * steps / dt, "unit": "frames/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "priming_steps": 4,
                     "bytes_over_pcie_per_step": 2 * 172 * self.S, "GB_per_s_each_way": 172 * self.S * steps / dt / 1e9,
                     "note": "wmx_pipe_submit / wmx_pipe_wait (the library's own C pipeline): pinned host rows, 3 slots in flight, copy-in / "
                             "copy-out streams beside the compute stream; the streaming steps fed the pinned slots' datagrams again, so "
                             "they are excluded from parity_checked"}
        return self.pcie

    def parity_check(self):
        from oracle import loader
        port = loader.port()
        n_res = max(k for k, _ in self.rec) + 1
        far = np.concatenate([self.far_host[(k % self.K) * 160:(k % self.K) * 160 + 160] for k in range(n_res)])
        worst, n = 0, 0
        for col, s in enumerate(self.sample):
            dgs = np.stack([self.base_dg[k % self.K, s % 256] for k in range(n_res)])
            want = loader.run_rtp_chain(port, far, dgs)
            for k, got in self.rec:
                g = got[col].cpu().numpy()
                assert np.array_equal(g[:12], want[k][:12]), "RTP header differs"
                worst, n = max(worst, int(np.abs(g[12:].astype(np.int16) - want[k][12:].astype(np.int16)).max())), n + 1
        return {"streams": len(self.sample), "datagrams_compared": n, "max_code_step": worst, "headers": "identical",
                "oracle": "orc_rtp_ingest -> orc_*.c chain -> orc_rtp_egress (port)", "steps_replayed": n_res}

    def cpu_baseline(self, budget_s):
        from oracle import loader
        port = loader.port()
        n = 400
        far = np.tile(self.far_host, n // self.K + 1)[: n * 160]
        dgs = np.tile(self.base_dg[:, 0], (n // self.K + 1, 1))[:n]
        reps, v1, nc, vn = _cpu_rates(lambda: loader.run_rtp_chain(port, far, dgs), 2 * n, budget_s)
        return {"value": v1, "unit": "frames/s", "cores": 1, "kind": "port", "all_cores_value": vn, "all_cores": nc, "cpu_model": _cpu_model(),
                "sample": "%d x %d datagrams of one stream through orc_rtp_ingest -> oracle chain -> orc_rtp_egress (python loop over "
                          "datagrams around the C calls), 1 thread; then all %d cores" % (reps, n, nc)}


class MfftWorkload:
    """math/fft.c's intended use (fft_stream's 1024-sample pool): one 1024-point real FFT (FFTR) with amplitude curve
    per stream per step.  Algorithmic bytes per transform = 4 096 in + 4 096 amplitude out = 8 192 B."""
    name = "mfft_fftr_1024_amplitude"
    pmc_tag = "mfft"
    dtype = "f32 data, f64 twiddles"
    bytes_per_frame = 8192.0
    dominant_kernel = "mfft_regs_kernel<1, false, 9>"
    dominant_bytes_per_frame = 8192.0
    N = 1024

    def __init__(self, dev, n_streams, rank):
        from wmix_amd import mfft
        self.mfft = mfft
        self.n_frames = n_streams
        g = torch.Generator(device="cpu").manual_seed(1234 + rank)
        self.x = (torch.randn((n_streams, self.N), generator=g) * 3000).to(dev)
        self.t = _StageTimer("fftr")

    def step(self, timed):
        self.out = self.t.run("fftr", timed, lambda: self.mfft.transform(1, self.x, None, want="a"))

    def parity_check(self):
        """64 sampled transforms of the last step against oracle/orc_mfft.c, the amplitude curve bit for bit (the transform is
        stateless: every step computes the same curves from the same input)."""
        from oracle import loader
        port = loader.port()
        rows = [int(i) for i in np.linspace(0, self.n_frames - 1, 64)]
        got = self.out["a"][rows].cpu().numpy()
        x = self.x[rows].cpu().numpy()
        bad, worst = 0, 0.0
        for i in range(len(rows)):
            want = loader.mfft(port, 1, x[i], None, self.N, prefix="orc", want="a")["a"]
            neq = got[i].view(np.uint32) != want.view(np.uint32)
            bad += int(neq.sum())
            if neq.any():
                worst = max(worst, float(np.abs(got[i] - want).max()))
        return {"transforms": len(rows), "values_compared": len(rows) * self.N, "values_differing": bad, "max_abs_diff": worst,
                "max_lsb": 0 if bad == 0 else 1, "oracle": "oracle/orc_mfft.c (port), float32 bit patterns"}

    def dominant_ms(self):
        return self.t.dominant_ms()

    def stage_ms(self):
        return {"fftr": self.t.mean_ms("fftr")}

    def config(self):
        return {"workload": self.name, "transforms_per_step_per_gpu": self.n_frames, "frame": "1024 x float32"}

    def cpu_baseline(self, budget_s):
        from oracle import loader
        port = loader.port()
        x = self.x[0].cpu().numpy()

        def one():
            for _ in range(200):
                loader.mfft(port, 1, x, None, self.N, prefix="orc", want="a")
        reps, v1, nc, vn = _cpu_rates(one, 200, budget_s)
        return {"value": v1, "unit": "frames/s", "cores": 1, "kind": "port", "all_cores_value": vn, "all_cores": nc, "cpu_model": _cpu_model(),
                "sample": "%d x 200 FFTR(1024) through oracle/orc_mfft.c (-O2), 1 thread; then all %d cores (python call overhead "
                          "included: one ctypes call per transform)" % (reps, nc)}


class ChainWorkload:
    """BASELINE.json configs[2]: the daemon's record chain NS -> AEC -> AGC -> VAD (src/wmix.c:613-709), 16 kHz mono,
    65536 streams per GPU sharing one far-end reference, one 10 ms packet per stream per step.  With N > 1 ranks the
    far-end packet is broadcast from rank 0 over RCCL every step; streams never talk to each other.
    Algorithmic bytes per stream-frame = 640 PCM + 2 x (12 200 + 11 700 + 668 + 736) state = 51 248 B;
    the dominant kernel is the AEC near-end kernel: 320 + 320 + 2 x 11 700 = 24 040 B per stream-frame."""
    name = "chain_ns_aec_agc_vad_16k_mono"
    dtype = "f32"
    bytes_per_frame = 51248.0
    dominant_kernel = "aec_near_kernel<2>"
    dominant_bytes_per_frame = 24040.0
    freq, pkt = 16000, 160
    with_agc_vad = True
    extra_stages = 0      # WMX_CHAIN_NSX | WMX_CHAIN_AECM for the fixed-point chain
    timer_dominant = "aec"

    def __init__(self, dev, n_streams, rank, dist=None, packets=1, interval_ms=10, cohorts=1, cohort_layout="arrival", coalesce=False,
                 far_ends=1, far_chunk=1):
        from wmix_amd import synth
        from wmix_amd.chain import AEC, AGC, NS, VAD, ChainBatch
        global broadcast_far
        from wmix_amd.shard import broadcast_far
        self.n_streams = n_streams
        self.n_frames = n_streams * packets  # 10 ms stream-frames per step
        self.dist = dist
        # SURVEY 8d recipe: shared far-end = LCG noise A=8000; near = far delayed 40 samples / 2 + noise A=200 + a
        # 3000 sin(0.01 t) tone gated on/off every 100 frames.  One full gate period (200 packets) of 256 distinct streams,
        # tiled over the batch on the device; a default run (256 priming + warm-up + timed + 16 steps) walks through the
        # tone-on, tone-off and both transitions, so the data-dependent branches (VAD decisions, NS feature updates, AEC
        # near-state) are not frozen on one 80 ms loop as in round 1.
        self.K = 200
        # --far-ends N: N DISTINCT far-end signals, stream s cancelled against far-end s * N // S (neighbours share one: a mix group,
        # a call) -- aec_process2 takes the far-end per handle (src/webrtc.c:410-483), and every mix group / call of a telephony
        # server is its own.  All N are created together (one tick, one delay): their control planes run in lockstep but their
        # far-end histories differ, so they never fold.  16 distinct far signals x 16 near-end streams each, tiled.
        self.far_ends = int(far_ends)
        assert 1 <= self.far_ends <= n_streams and (self.far_ends == 1 or (cohorts == 1 and dist is None)), \
            "--far-ends: one rank, no --cohorts (every far-end is a cohort of its own, created at step 0)"
        if self.far_ends > 1:
            FU = 16
            fars = np.stack([synth.far_end(3000 + 101 * u, self.K, self.pkt) for u in range(FU)])  # [FU, K * pkt]
            base = np.concatenate([synth.near_end(3001 + 7919 * rank + 997 * u, 16, self.K, self.pkt, far=fars[u]) for u in range(FU)])
            base = base.reshape(256, self.K, self.pkt)  # pattern u * 16 + v: near-end v of far signal u
            self.far_pat = fars.reshape(FU, self.K, self.pkt)
            sidx = np.arange(n_streams)
            self.far_of = sidx * self.far_ends // n_streams            # the far-end of stream s
            self.pat_of = (self.far_of % FU) * 16 + sidx % 16           # its near-end pattern
            far = fars[0]
        else:
            far = synth.far_end(3000, self.K, self.pkt)  # the same far-end on every rank (rank 0's copy is broadcast)
            base = synth.near_end(3001 + 7919 * rank, 256, self.K, self.pkt, far=far).reshape(256, self.K, self.pkt)
            self.pat_of = np.arange(n_streams) % 256
        self.base, self.far_host = base, far.reshape(self.K, self.pkt)
        b = torch.from_numpy(np.ascontiguousarray(base.transpose(1, 0, 2))).to(dev)  # [K, 256, pkt]
        self.inp = b[:, torch.from_numpy(self.pat_of).to(dev)]                         # [K, S, pkt] packet-major
        self.far_src = torch.from_numpy(far.reshape(self.K, self.pkt).copy()).to(dev)
        if self.far_ends > 1:  # [K, N, pkt]: far-end j hears far signal j % 16
            fp = torch.from_numpy(np.ascontiguousarray(self.far_pat.transpose(1, 0, 2))).to(dev)
            self.far_src = fp[:, torch.arange(self.far_ends, device=dev) % 16].contiguous()
        if dist is not None and rank != 0:
            self.far_src.zero_()  # only rank 0 has the far-end; the others hear it through the broadcast alone
        self.P = packets  # 10 ms packets per stream per step (1 = one packet per launch; 2 = the daemon's own 20 ms calls)
        assert self.K % self.P == 0
        # several GPUs: two far-end receive buffers; the packet of step k + 1 is broadcast while step k computes (the daemon
        # itself hands the AEC a far-end that is 400 ms old, src/wmix.c:651-657: it is known long before it is needed)
        # --far-chunk K: ONE broadcast carries the far-end of K steps ("one ncclBroadcast per batch of K frames"):
        # the collective's host cost -- the one thing that can bend the scaling curve at a 1 ms step -- is paid once per K steps
        self.FC = int(far_chunk)
        assert self.FC >= 1
        self.far = [torch.zeros(self.FC * self.P, self.pkt, dtype=torch.int16, device=dev) for _ in range(2)]
        self.far_work = [None, None]
        # interval_ms = 20 is the daemon's own cadence (WMIX_INTERVAL_MS, src/wmixConf.h:112: what it hands aec_init / agc_init /
        # vad_init, src/wmix.c:636, 684, 703): VAD packets of 20 ms, AEC packets of 20 ms at 8 kHz.  Such a packet must lie in
        # one piece, so the batch is tick-major then -- [tick][stream][P packets] -- instead of packet-major.
        self.interval_ms = interval_ms
        self.tick_major = interval_ms == 20
        if self.tick_major:
            assert self.P % 2 == 0, "--interval-ms 20 needs an even --packets-per-step (20 ms packets)"
            K, S, P = self.K, n_streams, self.P
            self.inp = self.inp.view(K // P, P, S, self.pkt).permute(0, 2, 1, 3).contiguous()  # [K / P, S, P, pkt]
            self.work = torch.empty_like(self.inp[0])
        else:
            self.work = torch.empty_like(self.inp[0:self.P])
        # the four stages behind ONE C call per step (wmx_chain_process, the heartbeat of src/wmix.c:613-709)
        self.chain = ChainBatch(n_streams, 1, self.freq, interval_ms, 5,  # volumeAgc default 5, src/wmix.c:1596
                                ((NS | AEC | AGC | VAD) if self.with_agc_vad else (NS | AEC)) | self.extra_stages,
                                n_cohorts=self.far_ends, stream_cohort=(self.far_of if self.far_ends > 1 else None))
        self.rank = rank
        self.t = _StageTimer(self.timer_dominant)
        self.k = 0
        self.near_ms, self.far_ms, self.aec_launches = 0.0, 0.0, 0
        self.sample = [int(i) for i in np.linspace(0, n_streams - 1, 16)]
        self.rec = []
        # --cohorts N: the streams are N groups of handles created at N distinct ticks (the reference makes a handle inside the
        # heartbeat on first use, src/wmix.c:617-618, 635-636): group j joins at step j -- wmx_chain_add_cohort (aec_init of the
        # shared part: a control plane and a far-end history of its own from that tick on), *_init of its members in every stage,
        # and the active mask grows.  "arrival": a group's streams are neighbours (slots handed out in arrival order);
        # "interleaved": stream s belongs to group s % N (slots scattered by churn: the streams of a workgroup hear different
        # cohorts' far-end histories).
        self.n_cohorts, self.cohort_layout = int(cohorts), cohort_layout
        # --coalesce: wmx_chain_coalesce behind every step -- cohorts whose control planes have converged (same delay, start-up over,
        # same phase of the re-blocking) are merged after a word-for-word comparison of their far-end slabs on the device; the
        # priming grows by the blocks that takes and the merge rounds (32 pairs per call)
        self.coalesce, self.merged = bool(coalesce) and int(cohorts) > 1, 0
        assert 1 <= self.n_cohorts <= n_streams
        if self.n_cohorts > 1:
            sidx = np.arange(n_streams)
            self.join_of = (sidx % self.n_cohorts) if cohort_layout == "interleaved" else (sidx * self.n_cohorts // n_streams)
            self.members = [np.flatnonzero(self.join_of == j).astype(np.int32) for j in range(self.n_cohorts)]
            self.active = np.zeros(n_streams, np.uint8)
            self.host_ctl_s = 0.0
        else:
            self.join_of = np.zeros(n_streams, np.int64)

    def min_prime(self):
        """untimed steps needed before every stream has joined and is past the start-up phases"""
        # blocks until a young cohort can fold: the float AEC's far-end rings filled and its far power rounded onto the older cohort's
        # (250 blocks and a margin); the AECM's binary far spectrum thresholds meeting the older cohort's bit for bit (~1 500 blocks)
        blocks = 1800 if getattr(self, "extra_stages", 0) & 32 else 500
        settle = (blocks * 64 // (self.pkt * self.P) + 120 + self.n_cohorts // 8) if self.coalesce else 0
        return self.n_cohorts - 1 + settle

    def _join(self, j):
        """group j's handles are created in front of step j"""
        if j == 0:
            c = 0  # the cohort the chain was created with
            self.chain.reset_cohort(0)
        else:
            c = self.chain.add_cohort()
        assert c == j or self.coalesce
        self.chain.reset_streams(self.members[j], cohort=c)
        self.active[self.members[j]] = 1
        self.chain.set_active(None if j == self.n_cohorts - 1 else self.active)

    def _far_for(self, step_index):
        """The far-end packets of a step.  One GPU: read where they lie.  Several: rank 0's packets arrive through the
        broadcast buffer of that step's parity, requested one step ahead."""
        P = self.P
        k = (step_index * P) % self.K
        if self.dist is None:
            return self.far_src[k:k + P]  # [P, pkt], or [P, N, pkt] with --far-ends N
        c, j = divmod(step_index, self.FC)  # chunk c of FC steps, step j inside it
        b = c & 1
        if j == 0:
            if self.far_work[b] is None:  # first chunk: nothing was requested ahead
                self._request_far(c)
            w, self.far_work[b] = self.far_work[b], None
            if w is not True:
                w.wait()
            self._request_far(c + 1)  # the next chunk travels while this one computes
        return self.far[b][j * P:(j + 1) * P]

    def _request_far(self, chunk_index):
        b = chunk_index & 1
        if self.rank == 0:
            n = self.FC * self.P
            k = (chunk_index * n) % self.K
            idx = (k + torch.arange(n, device=self.far_src.device)) % self.K  # a chunk may wrap around the K-packet pattern
            self.far[b].copy_(self.far_src[idx])
        self.far_work[b] = broadcast_far(self.far[b], self.dist, src=0, async_op=True) or True

    def timed_region(self, on):
        """Inside the timed region the library itself records HIP events around the AEC's kernels, on the launch stream
        (wmx_aec_set_timing): the dominant kernel's own duration, far kernel excluded."""
        if self.extra_stages & 32:  # AECM: no float AEC handle to time; the dominant kernel comes from the stage breakdown
            return
        self.chain.set_aec_timing(on)
        if on:
            self.chain.aec_host_ctl()  # start over
        else:
            n, f, r = self.chain.aec_timing()
            self.aec_launches, self.far_ms, self.near_ms = self.aec_launches + n, self.far_ms + f, self.near_ms + r
            nl, sec = self.chain.aec_host_ctl()
            self.host_ctl_us = sec / nl * 1e6 if nl else None

    def step(self, timed):
        P = self.P
        k = (self.k * P) % self.K
        step_index = self.k
        self.k += 1
        far = self._far_for(step_index)
        if step_index < self.n_cohorts and self.n_cohorts > 1:
            self._join(step_index)
        if self.coalesce and step_index > 0:
            self.merged += len(self.chain.coalesce(32))
        if self.tick_major:
            src = self.inp[k // P]
            if timed == "all" and self.far_ends == 1:
                for name, fn in self.chain.stage_calls_stream_major(far, src, self.work):
                    self.t.run(name, timed, fn)
            else:
                rc, _, _ = self.chain.process(far, src, out=self.work)
                assert rc == 0
        elif timed == "all" and self.far_ends == 1:
            for name, fn in self.chain.stage_calls_packet_major(far, self.inp[k:k + P], self.work):
                self.t.run(name, timed, fn)
        else:
            rc, _, _ = self.chain.process_packet_major(far, self.inp[k:k + P], out=self.work)
            assert rc == 0
        if timed is not True:
            # outside the timed region: keep what the sampled streams produced ([P, 16, pkt]), for parity_check()
            got = self.work[self.sample].transpose(0, 1) if self.tick_major else self.work[:, self.sample]
            self.rec.append((step_index, got.clone()))

    def dominant_ms(self):
        if self.extra_stages:
            return self.t.mean_ms(self.timer_dominant)
        return self.near_ms / self.aec_launches if self.aec_launches else None

    def stage_ms(self):
        d = {k: self.t.mean_ms(k) for k in ("ns", "aec", "agc", "vad") if self.t.mean_ms(k) is not None}
        if self.aec_launches:
            d["aec_far_kernel (timed region)"] = self.far_ms / self.aec_launches
            d["aec_near_kernel (timed region)"] = self.near_ms / self.aec_launches
        return d

    def parity_check(self):
        """Replays 16 sampled streams through the oracle chain for exactly the packets this run fed (priming, warm-up,
        timed and breakdown steps) and compares every packet recorded outside the timed region; the packets behind the
        timed region depend on every timed step through the filter / model state.  The float path is bit-exact since round 5
        (glibc's powf algorithm on the device); max_lsb is reported, the tests require 0."""
        from oracle import loader
        port = loader.port()
        T, P = self.k, self.P
        stages = 15 if self.with_agc_vad else 3
        worst, n, n_off = 0, 0, 0
        for col, s in enumerate(self.sample):
            t0 = int(self.join_of[s])  # the step in front of which this stream's handles were created (0 without --cohorts)
            far_host = self.far_pat[self.far_of[s] % 16] if self.far_ends > 1 else self.far_host  # the far-end THIS stream hears
            far = np.concatenate([far_host[(k * P + p) % self.K] for k in range(t0, T) for p in range(P)])
            near = np.concatenate([self.base[self.pat_of[s], (k * P + p) % self.K] for k in range(t0, T) for p in range(P)])
            # one oracle call per step of P packets, like wmx_chain_process: ns / aec / agc loop over the packets of a call, vad_process
            # analyses and attenuates the call's FIRST packet only -- the daemon's own 20 ms call is P = 2
            want = self._oracle_chain(loader, port, stages, far, near).reshape(T - t0, P, self.pkt)
            for k, got in self.rec:
                if k < t0:
                    continue
                d = np.abs(got[:, col].cpu().numpy().astype(np.int32) - want[k - t0].astype(np.int32))
                worst, n, n_off = max(worst, int(d.max())), n + P, n_off + int((d > 0).sum())
        return {"streams": len(self.sample), "packets_compared": n, "max_lsb": worst, "samples_off_by_one": n_off,
                "oracle": "oracle/orc_*.c chain (port), one run per sampled stream started at the stream's own join step",
                "steps_replayed": T}

    def _oracle_chain(self, loader, port, stages, far, near):
        return loader.run_chain(port, 1, self.freq, 5, stages, far, near, self.pkt * self.P, prefix="orc", interval_ms=self.interval_ms)

    def measure_pcie(self, steps):
        """The heartbeat's own boundary is a package in HOST memory (buffSrc, src/wmix.c:609-709).  The same workload through
        wmx_pipe_create_pcm / wmx_pipe_submit / wmx_pipe_wait: pinned rows in and out, H2D of step k + 1 and D2H of step k - 1 beside the
        compute of step k, a chain of its own (primed like the resident one).  Reported as `pcie_inclusive`; never `value`."""
        if (self.dist is not None or self.far_ends > 1 or self.n_cohorts > 1 or self.P != 1 or self.tick_major or self.extra_stages
                or os.environ.get("WMIX_BENCH_NO_PCIE") == "1"):
            return None
        from wmix_amd.chain import AEC, AGC, NS, VAD
        from wmix_amd.pipeline import PcmChain, StreamingPipe
        ch = PcmChain(self.n_streams, self.inp.device, 1, self.freq, 10, 5, (NS | AEC | AGC | VAD) if self.with_agc_vad else (NS | AEC), slots=3)
        pipe = StreamingPipe(ch)
        host = self.inp[: pipe.SLOTS].cpu().numpy()  # three consecutive packets of every stream; the slots are fed these again and again
        for sl in range(pipe.SLOTS):
            pipe.h_in[sl][:] = host[sl]
        far = self.far_src.view(self.K, 1, self.pkt)
        k = 0
        for _ in range(300):  # past the canceller's start-up
            pipe.submit(far[k % self.K])
            k += 1
        pipe.drain()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.submit(far[k % self.K])
            k += 1
        pipe.drain()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        row = self.pkt * 2
        self.pcie = {"value": self.n_frames * steps / dt, "unit": "frames/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "priming_steps": 300,
                     "bytes_over_pcie_per_step": 2 * row * self.n_streams, "GB_per_s_each_way": row * self.n_streams * steps / dt / 1e9,
                     "note": "wmx_pipe_create_pcm + wmx_pipe_submit / wmx_pipe_wait: pinned host rows of one 10 ms package per stream, 3 slots "
                             "in flight, copy-in / copy-out streams beside the compute stream; a second chain of the same shape, primed 300 "
                             "steps; its output is covered by tests/test_pipeline_gpu.py, not by parity_checked"}
        ch.close()
        return self.pcie

    def config(self):
        return {"workload": self.name, "streams_per_gpu": self.n_streams, "packets_per_stream_per_step": self.P,
                "pcie_inclusive": getattr(self, "pcie", None),
                "interval_ms": self.interval_ms,
                "cohorts": self.n_cohorts,
                "far_ends": self.far_ends,
                "far_end_device_bytes": self._far_bytes(),
                "coalesce": ({"cohorts_merged": self.merged, "cohorts_live": self.chain.live_cohorts(), "cohort_ids": self.chain.n_cohorts}
                             if self.coalesce else None),
                "aec_host_control_plane_us_per_launch": getattr(self, "host_ctl_us", None),
                "cohort_layout": (self.cohort_layout if self.n_cohorts > 1 else None),
                "layout": "tick-major [tick][stream][P x 10 ms]" if self.tick_major else "packet-major [packet][stream]",
                "frame": "%d x int16 (10 ms @ %d kHz mono)" % (self.pkt, self.freq // 1000),
                "input": "SURVEY 8d recipe: far = LCG noise A=8000; near = far delayed 40 / 2 + noise A=200 + 3000 sin(0.01 t) gated "
                         "every 100 frames; 256 distinct streams x %d packets, tiled" % self.K,
                "far_end": ("shared, RCCL broadcast from rank 0: one collective per %d step(s), the next chunk travels while this one "
                            "computes" % self.FC if self.dist is not None else
                            ("%d distinct far-ends, stream s hears far-end s * N // S; resident in HBM" % self.far_ends if self.far_ends > 1
                             else "shared, resident in HBM (one GPU: nothing to broadcast)")),
                "sum_order": "reference (the only one the library has)",
                "host_calls_per_step": "one: wmx_chain_process (NS, AEC far + near, AGC, VAD launched back to back by the C library)",
                "aec_launch": "far kernel + near kernel; roofline = the near kernel alone, timed by HIP events the library records "
                              "on the launch stream around it (wmx_aec_set_timing)"}

    def _far_bytes(self):
        """device bytes of the far-end slabs (one per far-end: the re-blocking ring of 250 partitions and the consumed-block history)"""
        from wmix_amd._lib import lib
        try:
            a = lib().wmx_chain_aec(self.chain._h)
            return int(self.far_ends * lib().wmx_aec_cohort_state_bytes(a)) if a else None
        except Exception:
            return None

    def cpu_baseline(self, budget_s):
        from oracle import loader
        port = loader.port()
        n = 2000
        far = np.tile(self.far_host.reshape(-1), n // self.K + 1)[: n * self.pkt]
        near = np.tile(np.ascontiguousarray(self.base[0].reshape(-1)), n // self.K + 1)[: n * self.pkt]
        stages = 15 if self.with_agc_vad else 3
        chain = ("ns_process -> aec_process2 -> agc_process -> vad_process" if self.with_agc_vad else "ns_process -> aec_process2")
        ref_lib = None
        if loader.have_ref():
            try:  # a prebuilt library that does not load on this host must not take the benchmark down
                ref_lib = loader.ref()
            except Exception:
                ref_lib = None
        if ref_lib is not None:
            # the real reference (src/webrtc.c over the vendored WebRTC, gcc -O2, generic-C AEC kernels), prebuilt by
            # oracle/Makefile where /root/reference exists; our restatement timed beside it for comparison
            reps, v1, nc, vn = _cpu_rates(lambda: loader.run_chain(ref_lib, 1, self.freq, 5, stages, far, near, self.pkt, prefix="ref"),
                                          n, budget_s * 0.75)
            _, vp, _, _ = _cpu_rates(lambda: loader.run_chain(port, 1, self.freq, 5, stages, far, near, self.pkt, prefix="orc"), n,
                                     budget_s * 0.25)
            return {"value": v1, "unit": "frames/s", "cores": 1, "kind": "reference", "all_cores_value": vn, "all_cores": nc,
                    "cpu_model": _cpu_model(), "port_value": vp,
                    "sample": "%d x %d packets of one %d kHz stream through the reference chain %s (oracle/_ref/libwmixref.so, "
                              "-O2), 1 thread; then one stream per thread on all %d cores" % (reps, n, self.freq // 1000, chain, nc)}
        reps, v1, nc, vn = _cpu_rates(lambda: loader.run_chain(port, 1, self.freq, 5, stages, far, near, self.pkt, prefix="orc"), n,
                                      budget_s)
        return {"value": v1, "unit": "frames/s", "cores": 1, "kind": "port", "all_cores_value": vn, "all_cores": nc,
                "cpu_model": _cpu_model(),
                "sample": "%d x %d packets of one %d kHz stream through the oracle chain (oracle/orc_*.c, -O2), 1 thread; then one "
                          "stream per thread on all %d cores" % (reps, n, self.freq // 1000, nc)}


class NsAgcMix32kWorkload:
    """BASELINE.json configs[4]: 2-channel 32 kHz NS + AGC per source, then wmix_load_data's resample to the 8 kHz mono ring
    with an 8-way saturating mix, 32 768 sources per GPU in 4 096 mix groups, and the play thread's 10 ms drain.
    Algorithmic bytes per source-frame = 2 560 PCM + 2 x (12 200 + 2 048 + 668) state + 160 x (1 + 1/8) ring = 32 550 B
    (SURVEY 8d); the NS kernel: 1 280 + 1 280 + 2 x (12 200 + 2 048) = 31 056 B."""
    name = "ns_agc_32k_2ch_mix8_to_8k"
    pmc_tag = "ns_agc_mix_32k"
    dtype = "f32 (NS), int16/int32 (AGC, mix)"
    bytes_per_frame = 32550.0
    dominant_kernel = "ns_kernel<256, 2>"
    dominant_bytes_per_frame = 31056.0
    N = 8  # sources per mix group

    def __init__(self, dev, n_streams, rank):
        from wmix_amd.agc import AgcBatch
        from wmix_amd.mix import MixBatch
        from wmix_amd.ns import NsBatch
        S = self.n_frames = n_streams
        assert S % self.N == 0
        self.K, per = 4, 640
        rng = np.random.default_rng(500 + rank)
        t = np.arange(self.K * 320)
        base = np.zeros((64, self.K * 320, 2), np.int16)
        for s in range(64):
            tone = 6000 * np.sin(2 * np.pi * (150 + 31 * s) * t / 32000)
            base[s, :, 0] = np.clip(tone + rng.integers(-2000, 2000, t.size), -32768, 32767)
            base[s, :, 1] = base[s, :, 0] // 3
        self.base = base.reshape(64, self.K, per)
        x = np.tile(self.base, (S // 64 + 1, 1, 1))[:S].transpose(1, 0, 2)
        self.inp = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        self.flat = torch.zeros(S * per + 2, dtype=torch.int16, device=dev)  # + the mixer's 2-sample look-ahead
        self.work = self.flat[: S * per].view(1, S, per)
        self.src = torch.as_strided(self.flat, (S // self.N, self.N, per + 2), (self.N * per, per, 1))
        self.ns, self.agc = NsBatch(S, 2, 32000), AgcBatch(S, 2, 32000, 5)
        self.mix = MixBatch(S // self.N, 1, 8000)
        self.t = _StageTimer("ns")
        self.k = 0
        # parity: 4 mix groups (their 32 sources) watched outside the timed region
        self.groups = sorted({int(g) for g in np.linspace(0, S // self.N - 1, 4)})
        self.watch = [g * self.N + i for g in self.groups for i in range(self.N)]
        self.rec = []

    def step(self, timed):
        k = self.k % self.K
        step_index = self.k
        self.k += 1
        S = self.n_frames
        self.t.run("ns", timed, lambda: self.ns.process_packet_major(self.inp[k:k + 1], self.work))
        self.t.run("agc", timed, lambda: self.agc.process(self.work[0].view(S, 2, 320)))  # 5 ms AGC packets at 32 kHz

        def mix():
            self.mix.set(0, 0, 1)
            self.mix.load(self.src, 1280, 32000, 2)
            self.mix.set(3200, 0, 1)
            return self.mix.drain(160)
        out = self.t.run("mix", timed, mix)
        if timed is not True:
            self.rec.append((step_index, self.work[0, self.watch].clone(), out[self.groups].clone()))

    def dominant_ms(self):
        return self.t.dominant_ms()

    def stage_ms(self):
        return {k: self.t.mean_ms(k) for k in ("ns", "agc", "mix")}

    def parity_check(self):
        """The 32 sources of 4 sampled mix groups replayed through the oracle's NS + AGC (2 x 32 kHz, R channel = high band,
        5 ms AGC packets) for exactly the packets fed; every packet recorded outside the timed region is compared, and so is the
        group's drained 10 ms of the 8 kHz ring against oracle/orc_mix.c fed with the ORACLE's source packets in call order
        (saturating accumulate).  NS float path: bit-exact; AGC and mix are integer."""
        from oracle import loader
        port = loader.port()
        per, N = 640, self.N
        want = {}
        z = np.zeros(self.k * per, np.int16)
        for s in self.watch:
            x = np.concatenate([self.base[s % 64, k % self.K] for k in range(self.k)])
            want[s] = loader.run_chain(port, 2, 32000, 5, 1 | 4, z, x, 320, prefix="orc").reshape(self.k, per)
        worst, worst_mix, n, n_mix = 0, 0, 0, 0
        for k, pcm, drained in self.rec:
            pcm, drained = pcm.cpu().numpy(), drained.cpu().numpy()
            for col, s in enumerate(self.watch):
                worst, n = max(worst, int(np.abs(pcm[col].astype(np.int32) - want[s][k].astype(np.int32)).max())), n + 1
            for gi, g in enumerate(self.groups):
                flat = np.concatenate([want[g * N + i][k] for i in range(N)] + [np.zeros(2, np.int16)])
                ring, _ = loader.mix_load(port, 1, 8000, 32000, 2, 1, 1, N, per * 2, 0, flat)
                worst_mix = max(worst_mix, int(np.abs(drained[gi].astype(np.int32) - ring[1600:1680].astype(np.int32)).max()))
                n_mix += 1
        return {"sources": len(self.watch), "packets_compared": n, "max_lsb": max(worst, worst_mix), "max_lsb_ns_agc": worst,
                "mix_groups": len(self.groups), "drains_compared": n_mix, "max_lsb_mix": worst_mix,
                "oracle": "oracle/orc_ns.c + orc_agc.c per source, orc_mix.c per group (port)", "steps_replayed": self.k}

    def config(self):
        return {"workload": self.name, "sources_per_gpu": self.n_frames, "mix_groups": self.n_frames // self.N,
                "frame": "640 x int16 (10 ms @ 32 kHz, 2 channels) -> 80 x int16 (10 ms @ 8 kHz mono) per group"}

    def cpu_baseline(self, budget_s):
        from oracle import loader
        port = loader.port()
        n = 1000
        x = np.tile(np.ascontiguousarray(self.inp[:, 0].cpu().numpy().reshape(-1)), n // self.K + 1)[: n * 640]
        z = np.zeros_like(x)
        reps, v1, nc, vn = _cpu_rates(lambda: loader.run_chain(port, 2, 32000, 5, 1 | 4, z, x, 320, prefix="orc"), n, budget_s)
        return {"value": v1, "unit": "frames/s", "cores": 1, "kind": "port", "all_cores_value": vn, "all_cores": nc, "cpu_model": _cpu_model(),
                "sample": "%d x %d packets of one 2 x 32 kHz source through NS + AGC of the oracle (mix excluded), 1 thread; then all %d "
                          "cores" % (reps, n, nc)}


class ConferenceWorkload:
    """The daemon's tick end to end, 4 096 daemons side by side (round-4 VERDICT "next" 3; BASELINE configs[4]'s shape on the play
    side, the shipped platform's format on the record side): per 20 ms tick and mix group eight 2-channel 32 kHz sources ->
    wmix_load_data into the group's 1 x 8000 ring -> the play thread's package -> playPkgBuff_add -> playPkgBuff_get(400 ms) = the
    far-end of THAT group's eight record streams -> NS -> AEC -> AGC -> VAD (20 ms AEC and VAD packets) -> wmix_pcm_zoom to 1 x 8000
    (src/wmix.c:1347-1440 with :528-780 inside; one C call per half: wmx_tick_play / wmx_tick_record).  One mix group = one control
    cohort with a far-end of its own: nothing folds, 4 096 far-end histories are live.  A step = one tick; `value` counts the RECORD
    side's 10 ms stream-frames (2 per stream and tick); the play side's 32 768 source packages per tick ride along.
    Algorithmic bytes per record stream-frame: 320 PCM + 2 x (6 000 + 11 700 + 668 + 736) state (SURVEY 8d, 8 kHz) + the group's far-end
    spectra, which its R = 8 streams share: 2 x (6 240 + 6 240) / 8 (SURVEY 8d "non-shared far adds xfBuf + xfwBuf") = 41 648 B, plus
    per source-frame 1 280 B of PCM in and the ring's 160 x (1 + 1/8) B = 1 460 B: 43 108 B.  Dominant kernel aec_near_kernel<1>:
    160 + 160 + 2 x 11 700 + 24 960 / 8 = 26 840 B."""
    name = "conference_mix8_32k_to_8k_chain"
    pmc_tag = "conference"
    dtype = "int16 (mix, FIFO, AGC, VAD), f32 (NS, AEC)"
    bytes_per_frame = 43108.0
    dominant_kernel = "aec_near_kernel<1>"
    dominant_bytes_per_frame = 26840.0
    N, R = 8, 8  # sources and record streams per mix group

    def __init__(self, dev, n_streams, rank):
        from wmix_amd import synth
        from wmix_amd.tick import TickBatch
        S = self.S = n_streams
        assert S % self.R == 0
        G = self.G = S // self.R
        self.n_frames = 2 * S  # 10 ms record stream-frames per tick
        self.K, U = 50, 16     # 50 ticks (1 s) of 16 distinct groups, tiled over the batch
        self.U = U
        per = 1280
        groups = [synth.conference_inputs(7000 + 7919 * rank + 13 * u, self.K, self.N, self.R, 32000, 2, loud=9000) for u in range(U)]
        self.src_host = np.stack([g[0] for g in groups])    # [U, K, N, per]
        self.loc_host = np.stack([g[1] for g in groups])    # [U, K, R, 160]
        gi = torch.arange(G, device=dev) % U
        s_pad = torch.zeros((self.K, U, self.N, per + 2), dtype=torch.int16, device=dev)  # + the mixer's look-ahead frame
        s_pad[..., :per] = torch.from_numpy(np.ascontiguousarray(self.src_host.transpose(1, 0, 2, 3))).to(dev)
        self.src = s_pad[:, gi].contiguous()                                               # [K, G, N, per + 2]
        loc = torch.from_numpy(np.ascontiguousarray(self.loc_host.transpose(1, 0, 2, 3))).to(dev)  # [K, U, R, 160]
        self.loc = loc[:, gi].reshape(self.K, S, 160).contiguous()
        self.tb = TickBatch(G, self.R)  # 1 x 8000, 20 ms, AEC_INTERVALMS 400, volumeAgc 5, the whole heartbeat
        self.rec = torch.zeros((S, 160), dtype=torch.int16, device=dev)
        self.zoom = torch.zeros((S, 160), dtype=torch.int16, device=dev)
        self.play = torch.zeros((G, 160), dtype=torch.int16, device=dev)
        self.line = torch.zeros((G, 320), dtype=torch.int16, device=dev)   # [previous far-end package | this one] of every group
        self.echo = torch.zeros((G, 160), dtype=torch.int16, device=dev)
        assert int(np.abs(self.loc_host.astype(np.int32)).max()) + 16384 <= 32767, "the room's saturating add must be a plain one here"
        self.t = _StageTimer("none")  # the dominant kernel is timed by the library's own events (wmx_aec_set_timing)
        self.k = 0
        self.near_ms, self.far_ms, self.aec_launches = 0.0, 0.0, 0
        self.groups = sorted({int(g) for g in np.linspace(0, G - 1, 4)})
        self.rows = [g * self.R + r for g in self.groups for r in range(self.R)]
        self.rec_log = []

    def _room(self, local, far):
        """near = sat(local + (the group's far-end delayed by 40 samples) >> 1): the loudspeaker in the microphone (the harness' input
        model, oracle.loader.tick_room on the device; it needs this tick's far-end, so it runs between the tick's two halves).  Four
        small launches: the delay line moves on, the shift, one broadcast add straight into the record rows.  The saturation can
        never act here (|local| <= 3 200 by construction, |echo| <= 16 384; checked once at start-up), so the add is plain int16."""
        N = 160
        self.line[:, :N].copy_(self.line[:, N:])
        self.line[:, N:].copy_(far)
        torch.bitwise_right_shift(self.line[:, N - 40: 2 * N - 40], 1, out=self.echo)
        torch.add(local.view(self.G, self.R, N), self.echo.view(self.G, 1, N), out=self.rec.view(self.G, self.R, N))

    def timed_region(self, on):
        L = self.tb
        from wmix_amd._lib import check, lib
        import ctypes as C_
        aec = lib().wmx_chain_aec(L.chain_handle())
        check(lib().wmx_aec_set_timing(aec, 1 if on else 0), "wmx_aec_set_timing")
        if not on:
            n, f, r = C_.c_int(0), C_.c_double(0), C_.c_double(0)
            check(lib().wmx_aec_timing(aec, C_.byref(n), C_.byref(f), C_.byref(r)), "wmx_aec_timing")
            self.aec_launches, self.far_ms, self.near_ms = self.aec_launches + n.value, self.far_ms + f.value, self.near_ms + r.value

    def step(self, timed):
        k = self.k % self.K
        step_index = self.k
        self.k += 1
        self.t.run("load (8 sources per group)", timed, lambda: self.tb.load(self.src[k], 2560, 32000, 2))
        far = self.t.run("play (drain, fifo add / get)", timed, lambda: self.tb.play(self.play))
        self.t.run("room (harness)", timed, lambda: self._room(self.loc[k], far))
        self.t.run("record", timed, lambda: self.tb.record(self.rec, self.zoom))
        if timed is not True:
            self.rec_log.append((step_index, self.play[self.groups].clone(), far[self.groups].clone(), self.zoom[self.rows].clone()))

    def dominant_ms(self):
        return self.near_ms / self.aec_launches if self.aec_launches else None

    def stage_ms(self):
        d = {k: self.t.mean_ms(k) for k in ("load (8 sources per group)", "play (drain, fifo add / get)", "room (harness)", "record")}
        if self.aec_launches:
            d["aec_far_kernel (timed region)"] = self.far_ms / self.aec_launches
            d["aec_near_kernel (timed region)"] = self.near_ms / self.aec_launches
        return d

    def config(self):
        return {"workload": self.name, "record_streams_per_gpu": self.S, "sources_per_gpu": self.G * self.N, "mix_groups": self.G,
                "far_ends": self.G, "cohorts": self.G, "tick_ms": 20,
                "frame": "record: 160 x int16 (20 ms @ 8 kHz mono) per stream and tick; sources: 1 280 x int16 (20 ms @ 32 kHz, 2 channels)",
                "far_end": "per mix group: that group's own playback out of the 400 ms FIFO, never shared between groups",
                "input": "16 distinct groups x 50 ticks (tone per source gated every 200 ms, talkers one second on / off), tiled; the "
                         "microphone signal = local + the group's far-end delayed 40 samples / 2, computed on the device between the "
                         "tick's two halves (in the timed step, not library code)"}

    def parity_check(self):
        """4 sampled mix groups as 4 daemons through the restatement (oracle.loader.tick_port: orc_load_data per source and tick, the
        package drain, orc_pkgfifo, the room, the oracle chain per record stream, orc_pcm_zoom) for exactly the ticks fed; every tick
        recorded outside the timed region is compared bit for bit: played package, far-end and the record streams' 1 x 8000 output."""
        from oracle import loader
        port = loader.port()
        T = self.k
        worst_i, worst, n, n_off = 0, 0, 0, 0
        for gi, g in enumerate(self.groups):
            u = g % self.U
            ticks = [t % self.K for t in range(T)]
            want = loader.tick_port(port, self.src_host[u][ticks], self.loc_host[u][ticks], 32000, 2)
            for t, play, far, zoom in self.rec_log:
                worst_i = max(worst_i, int(np.abs(play[gi].cpu().numpy().astype(np.int32) - want["play"][t]).max()),
                              int(np.abs(far[gi].cpu().numpy().astype(np.int32) - want["far"][t]).max()))
                d = np.abs(zoom[gi * self.R:(gi + 1) * self.R].cpu().numpy().astype(np.int32) - want["zoom"][t].astype(np.int32))
                worst, n, n_off = max(worst, int(d.max())), n + 2 * self.R, n_off + int((d > 0).sum())
        return {"mix_groups": len(self.groups), "record_streams": len(self.rows), "packets_compared": n, "max_lsb": max(worst, worst_i),
                "max_lsb_play_and_far_end": worst_i, "max_lsb_record": worst, "samples_off_by_one": n_off,
                "oracle": "oracle.loader.tick_port: one daemon per sampled group (orc_mix, orc_pkgfifo, orc_* chain, orc_pcm_zoom)",
                "steps_replayed": T}

    def cpu_baseline(self, budget_s):
        from oracle import loader
        port = loader.port()
        n = 100
        ticks = [t % self.K for t in range(n)]
        src, loc = self.src_host[0][ticks], self.loc_host[0][ticks]
        reps, v1, nc, vn = _cpu_rates(lambda: loader.tick_port(port, src, loc, 32000, 2), 2 * self.R * n, budget_s)
        return {"value": v1, "unit": "frames/s", "cores": 1, "kind": "port", "all_cores_value": vn, "all_cores": nc, "cpu_model": _cpu_model(),
                "sample": "%d x %d ticks of one daemon (8 sources, 8 record streams) through oracle.loader.tick_port (oracle/orc_*.c, -O2; "
                          "the per-tick mixer calls go through python), 1 thread; then one daemon per thread on all %d cores" % (reps, n, nc)}


class NsAec8kWorkload(ChainWorkload):
    """BASELINE.json configs[3]: NS -> AEC, 8 kHz mono, shared far-end, 131 072 streams per GPU (the 1 M streams of the
    config over 8 GPUs).  Algorithmic bytes per stream-frame = 320 PCM + 2 x (6 000 + 11 700) = 35 720 B; the
    AEC near-end kernel: 160 + 160 + 2 x 11 700 = 23 720 B."""
    name = "ns_aec_8k_mono"
    pmc_tag = "ns_aec_8k"
    bytes_per_frame = 35720.0
    dominant_kernel = "aec_near_kernel<1>"
    dominant_bytes_per_frame = 23720.0
    freq, pkt = 8000, 80
    with_agc_vad = False


class ChainFxWorkload(ChainWorkload):
    """The record chain of the reference's fixed-point builds (MAKE_WEBRTC_NSX, src/webrtc.c:512-521, and the AECM switch,
    src/webrtc.c:168-191): NSX -> AECM -> AGC -> VAD, 16 kHz mono, one wmx_chain_process call per step.  Integer end to end: the
    parity replay must be bit-exact.  Algorithmic bytes per stream-frame = 640 PCM + 2 x (5 516 + 3 900 + 668 + 736) = 22 280 B
    (DESIGN 7e / 7f state figures); dominant kernel nsx_kernel: 320 + 320 + 2 x 5 516 = 11 672 B."""
    name = "chain_nsx_aecm_agc_vad_16k_mono"
    pmc_tag = "chain_fx"
    dtype = "int16 / int32"
    bytes_per_frame = 22280.0
    dominant_kernel = "nsx_kernel<256, 1>"
    dominant_bytes_per_frame = 11672.0
    extra_stages = 16 | 32
    timer_dominant = "ns"

    def _oracle_chain(self, loader, port, stages, far, near):
        per = self.pkt * self.P
        x = loader.run_nsx(port, 1, self.freq, near, per, prefix="orc")
        x = loader.run_aecm(port, 1, self.freq, self.interval_ms, far, x, per, 0, prefix="orc")
        x = loader.run_agc(port, 1, self.freq, 5, x, per, prefix="orc")
        return loader.run_vad(port, 1, self.freq, self.interval_ms, x, per, prefix="orc")

    def cpu_baseline(self, budget_s):
        from oracle import loader
        port = loader.port()
        n = 2000
        far = np.tile(self.far_host.reshape(-1), n // self.K + 1)[: n * self.pkt]
        near = np.tile(np.ascontiguousarray(self.base[0].reshape(-1)), n // self.K + 1)[: n * self.pkt]
        P, self.P = self.P, 1
        try:
            reps, v1, nc, vn = _cpu_rates(lambda: self._oracle_chain(loader, port, 15, far, near), n, budget_s)
        finally:
            self.P = P
        return {"value": v1, "unit": "frames/s", "cores": 1, "kind": "port", "all_cores_value": vn, "all_cores": nc, "cpu_model": _cpu_model(),
                "sample": "%d x %d packets of one 16 kHz stream through the oracle's NSX, AECM, AGC and VAD (oracle/orc_*.c, -O2), 1 thread; "
                          "then one stream per thread on all %d cores" % (reps, n, nc)}


class Chain8kWorkload(ChainWorkload):
    """The format every platform of the reference ships with (PLAT_CHN 1, PLAT_FREQ 8000: platform/{alsa,hi3516,t31}/plat.h): the
    whole record chain NS -> AEC -> AGC -> VAD at 8 kHz mono.  With --interval-ms 20 --packets-per-step 2 it is the daemon's own
    heartbeat (20 ms AEC and VAD packets, src/webrtc.c:57-66, 239-248).  Algorithmic bytes per stream-frame = 320 PCM +
    2 x (6 000 + 11 700 + 668 + 736) = 38 528 B (SURVEY 8d figures); the AEC near-end kernel: 160 + 160 + 2 x 11 700 = 23 720 B."""
    name = "chain_ns_aec_agc_vad_8k_mono"
    pmc_tag = "chain_8k"
    bytes_per_frame = 38528.0
    dominant_kernel = "aec_near_kernel<1>"
    dominant_bytes_per_frame = 23720.0
    freq, pkt = 8000, 80
    with_agc_vad = True


WORKLOADS = {"chain_8k": (Chain8kWorkload, 131072), "chain_fx": (ChainFxWorkload, 65536), "g711": (G711Workload, 1 << 20), "ns": (NsWorkload, 4096), "nsx": (NsxWorkload, 65536), "aecm": (AecmWorkload, 65536), "rtp_chain": (RtpChainWorkload, 65536), "chain": (ChainWorkload, 65536),
             "mfft": (MfftWorkload, 65536), "ns_aec_8k": (NsAec8kWorkload, 131072),
             "ns_agc_mix_32k": (NsAgcMix32kWorkload, 32768), "conference": (ConferenceWorkload, 32768)}
DEFAULT_WORKLOAD = "chain"


# ----------------------------------------------------------------------------- driver
class StubCpuWorkload:
    """No GPU, no HIP library: exists so that the N > 1 launcher, the rendezvous, the far-end broadcast and the JSON
    contract can be driven on a CPU-only box over gloo (tests/test_bench_launcher.py).  Never a measurement."""
    name = "stub_cpu"
    dtype = "int16"
    bytes_per_frame = 640.0
    dominant_kernel = "none"
    dominant_bytes_per_frame = 640.0
    needs_gpu = False

    def __init__(self, dev, n_streams, rank, dist=None, packets=1, stream_offset=0, far_chunk=1):
        from wmix_amd.shard import broadcast_far
        if os.environ.get("WMIX_STUB_CHATTER") == "1" and rank > 0:  # tests: what ranks > 0 print is kept by the launcher
            print("stub rank %d says hello" % rank, flush=True)
        if os.environ.get("WMIX_STUB_FAIL_RANK") == str(rank):  # tests: one rank dies after the rendezvous
            sys.exit(7)
        self._bcast = broadcast_far
        self.n_frames = n_streams
        self.dist, self.rank = dist, rank
        self.lo = stream_offset  # global id of this rank's first stream (--total-streams: contiguous ranges, remainder on the first ranks)
        self.FC = int(far_chunk)  # --far-chunk K: one broadcast per K steps
        self.far = torch.zeros(self.FC, 160, dtype=torch.int16)
        self.n_bcast = 0
        self.acc = torch.zeros(n_streams, 160, dtype=torch.int32)
        self.gid = (torch.arange(n_streams, dtype=torch.int32) + self.lo) % 5  # what a stream adds depends on its GLOBAL id
        self.k = 0

    def step(self, timed):
        j = self.k % self.FC
        if j == 0:  # rank 0 knows the far-end of the next FC steps (their values: k + 1 .. k + FC) and sends them at once
            if self.rank == 0:
                self.far.copy_((self.k + 1 + torch.arange(self.FC, dtype=torch.int16))[:, None].expand(self.FC, 160))
            self._bcast(self.far, self.dist, src=0)
            self.n_bcast += 1
        self.k += 1
        self.acc += self.far[j].to(torch.int32)[None, :] + self.gid[:, None]

    def dominant_ms(self):
        return None

    def config(self):
        return {"workload": self.name, "streams_per_gpu": self.n_frames, "far_chunk": self.FC, "broadcasts": self.n_bcast,
                "far_sum": int((self.acc[0, 0] - self.k * self.gid[0]).item()) if self.n_frames else None}

    def parity_check(self):
        """every stream of this rank heard every far-end packet and knows its own global id (the `oracle` is arithmetic)"""
        want = self.k * (self.k + 1) // 2 + self.k * self.gid.to(torch.int64)
        ok = bool((self.acc.to(torch.int64) == want[:, None]).all().item())
        return {"streams": self.n_frames, "range": [self.lo, self.lo + self.n_frames], "packets_compared": self.k * self.n_frames,
                "max_lsb": 0 if ok else 1 << 15, "steps_replayed": self.k}

    def cpu_baseline(self, budget_s):
        return None


WORKLOADS["stub_cpu"] = (StubCpuWorkload, 4)


# ----------------------------------------------------------------------------- paced operation (real-time capacity)
PACED_KINDS = {"pcm16k": ("pcm", 16000), "pcm8k": ("pcm", 8000), "rtp8k": ("rtp", 8000)}


def paced_pattern(kind, slots, interval_ms=20, n_pattern=256, seed=7000, n_far=1):
    """The rows a paced run works on: `slots` consecutive ticks of n_pattern distinct streams (SURVEY 8d recipe: near = echo of the shared
    far-end + noise + gated tone), which the slots hold for the whole run -- tick t works on pattern slot t % slots.  Returns
    (far int16 [slots, far_samples], rows [slots, n_pattern, row]): int16 packages for "pcm", uint8 RTP/PCMA datagrams for "rtp"
    (header v=2 m=1 pt=8, A-law of the near-end: encoded by the library's own G.711 kernel, pinned exhaustively elsewhere)."""
    from wmix_amd import synth
    form, freq = PACED_KINDS[kind]
    pkt, ppc = freq // 100, (interval_ms // 10 if form == "pcm" else 2)
    if n_far > 1:  # a far-end per stream ("calls"): n_far distinct far signals, near row r is the echo of far signal r % n_far
        assert form == "pcm" and n_pattern % n_far == 0
        fars = np.stack([synth.far_end(seed + 101 * u, slots * ppc, pkt) for u in range(n_far)])
        near = np.stack([synth.near_end(seed + 1 + 7919 * r, 1, slots * ppc, pkt, far=fars[r % n_far])[0] for r in range(n_pattern)])
        pcm = np.ascontiguousarray(near.reshape(n_pattern, slots, ppc * pkt).transpose(1, 0, 2))
        return np.ascontiguousarray(fars.reshape(n_far, slots, ppc * pkt).transpose(1, 0, 2)), pcm  # far [slots, n_far, package]
    far = synth.far_end(seed, slots * ppc, pkt)
    near = synth.near_end(seed + 1, n_pattern, slots * ppc, pkt, far=far).reshape(n_pattern, slots, ppc * pkt)
    pcm = np.ascontiguousarray(near.transpose(1, 0, 2))  # [slots, n_pattern, package]
    far = far.reshape(slots, ppc * pkt)
    if form == "pcm":
        return far, pcm
    from wmix_amd.g711 import encode
    codes = encode("a", torch.from_numpy(pcm.reshape(-1).copy()).cuda()).cpu().numpy().reshape(slots, n_pattern, 160)
    pk = np.zeros((slots, n_pattern, 172), np.uint8)
    pk[:, :, 0], pk[:, :, 1] = 0x80, 0x88
    pk[:, :, 3] = np.arange(slots, dtype=np.uint8)[:, None]
    pk[:, :, 12:] = codes
    return far, pk


def paced_replay(kind, far, rows, pattern_row, n_ticks, interval_ms=20):
    """What the oracle says one stream's rows are after each of n_ticks ticks (tick t works on pattern slot t % slots): [n_ticks, row]."""
    from oracle import loader
    port = loader.port()
    form, freq = PACED_KINDS[kind]
    slots = rows.shape[0]
    t = np.arange(n_ticks) % slots
    far_seq = np.ascontiguousarray(far[t] if far.ndim == 2 else far[t, pattern_row % far.shape[1]]).reshape(-1)
    if form == "pcm":
        near = np.ascontiguousarray(rows[t, pattern_row]).reshape(-1)
        pkg = freq // 100 * (interval_ms // 10)
        return loader.run_chain(port, 1, freq, 5, 15, far_seq, near, pkg, prefix="orc", interval_ms=interval_ms).reshape(n_ticks, -1)
    return loader.run_rtp_chain(port, far_seq, np.ascontiguousarray(rows[t, pattern_row]))


def run_paced(dev, kind, S, tick_ms, ticks, sub=32768, slots=4, prime=150, resident=False, keep=24, interval_ms=None, parity=True, compute_streams=1,
              phases=1, calls=False):
    """One paced run: S concurrent streams, every stream's package due every tick_ms.  phases = 1: all S at the same instant, one tick
    through wmx_rt_tick (H2D, NS -> AEC -> AGC -> VAD, D2H; sub-batches overlapped) or, resident, wmx_rt_step_resident + a
    synchronisation.  phases = P > 1: P groups of S / P streams released tick_ms / P apart (wmx_rt_submit at the release, completion
    seen by polling), each with the whole period as its own.  Latency = scheduled release -> last row of the group in host memory
    (resident: -> its launches done).  The budget is the reference's own: tick_ms - 2 ms (src/wmix.c:536-538, 820).  16 sampled streams
    are replayed through the oracle for every tick of the run, start-up included, and compared on the last `keep` ticks."""
    import gc
    from wmix_amd.realtime import GpuClock, RtBatch, latency_summary, paced_groups, paced_loop
    form, freq = PACED_KINDS[kind]
    interval_ms = interval_ms or int(tick_ms)
    assert form == "pcm" or interval_ms == 20, "the RTP edge is 20 ms datagrams"
    t_start = time.perf_counter()
    # calls: every stream hears a far-end of its own (wmx_rt_create_pcm_calls; aec_process2's far-end is per handle): 16 distinct far signals
    far, rows = paced_pattern(kind, slots, interval_ms, n_far=16 if calls else 1)
    n_pattern = rows.shape[1]
    P = int(phases)
    bounds = [S * g // P for g in range(P + 1)]  # group g = streams [bounds[g], bounds[g + 1])
    rts = [RtBatch(bounds[g + 1] - bounds[g], dev, sub_batch=sub, slots=slots, kind=form, chn=1, freq=freq, interval_ms=interval_ms,
                   compute_streams=compute_streams, far_rows=calls) for g in range(P)]
    rt0 = rts[0]
    pat_of = np.arange(S) % n_pattern
    sample = sorted(set(int(i) for i in np.linspace(0, S - 1, 16)))
    sample_of = [[s - bounds[g] for s in sample if bounds[g] <= s < bounds[g + 1]] for g in range(P)]
    col_of = [[c for c, s in enumerate(sample) if bounds[g] <= s < bounds[g + 1]] for g in range(P)]
    if resident:
        dfar = None if calls else torch.from_numpy(far.reshape(slots, rt0.ppc, rt0.pkt10).copy()).to(dev)
        src, work, outb, evs, dfar_g = [], [], [], [], []
        for g in range(P):
            idx = torch.from_numpy(pat_of[bounds[g]:bounds[g + 1]]).to(dev)
            if calls:  # [slots][n_g, package]: stream s hears far signal (s % n_pattern) % 16
                dfar_g.append([torch.from_numpy(far[j]).to(dev)[idx % far.shape[1]].contiguous() for j in range(slots)])
            src.append([torch.from_numpy(rows[j]).to(dev)[idx].contiguous() for j in range(slots)])  # [n_g, row] per slot
            work.append(src[g][0].clone())
            outb.append(torch.empty_like(work[g]) if form == "rtp" else None)
            evs.append(torch.cuda.Event())
    else:
        for g, rt in enumerate(rts):
            for j in range(slots):
                if calls:
                    rt.fill_far(j, far[j][pat_of[bounds[g]:bounds[g + 1]] % far.shape[1]])
                else:
                    rt.h_far[j][:] = far[j].reshape(rt.far_shape)
                rt.fill(j, rows[j][pat_of[bounds[g]:bounds[g + 1]]])
    kept = np.zeros((keep, len(sample), rt0.row), rt0.row_dtype)
    t_of = [0] * P      # ticks group g has been through
    gpu_ms = []

    def submit(g):
        j = t_of[g] % slots
        if resident:
            rts[g].step_resident(work[g], dfar_g[g][j] if calls else dfar[j], outb[g])
            evs[g].record()
        else:
            assert rts[g].submit(None) == j
        t_of[g] += 1

    def poll(g):
        return evs[g].query() if resident else rts[g].poll()

    def wait(g):
        if resident:
            evs[g].synchronize()
        else:
            rts[g].wait()

    def after(k_paced, g):
        """behind the clock: keep the sampled rows of the last ticks; resident: put the next tick's input where the chain works in place"""
        j = (t_of[g] - 1) % slots
        if k_paced is not None and k_paced // P >= ticks - keep and sample_of[g]:
            if resident:
                res = (outb[g] if form == "rtp" else work[g])[sample_of[g]].cpu().numpy()
            else:
                res = rts[g].gather(j, sample_of[g])
            kept[k_paced // P - (ticks - keep), col_of[g]] = res
        if resident:
            work[g].copy_(src[g][t_of[g] % slots])

    def one(k_paced):  # phases = 1: the blocking tick
        if resident and k_paced is not None and k_paced % 8 == 0:  # every eighth tick between two events: the device's own time for it
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            submit(0)
            e1.record()
            e1.synchronize()
            gpu_ms.append(e0.elapsed_time(e1))
        else:
            submit(0)
            wait(0)

    for _ in range(prime):  # past the start-up phases of every stage, back to back
        for g in range(P):
            submit(g)
            wait(g)
            after(None, g)
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    try:
        if P == 1:
            lat, lag, clk = paced_loop(one, tick_ms, ticks, GpuClock(), after=lambda k: (after(k, 0), torch.cuda.current_stream().synchronize()))
        else:
            lat, lag, clk = paced_groups(submit, poll, wait, P, tick_ms, ticks, GpuClock(), after=after)
    finally:
        gc.enable()
    torch.cuda.synchronize()
    failed = sum(rt.failed_steps() for rt in rts)
    n_sub = sum(rt.B for rt in rts)
    for rt in rts:
        rt.close()
    out_d = latency_summary(lat, lag, tick_ms, clk)
    out_d.update({"kind": kind, "streams": S, "far_end_per_stream": bool(calls), "phases": P, "release": ("all %d streams at the same instant" % S) if P == 1 else
                  ("%d groups of %d streams, %.3g ms apart" % (P, S // P, tick_ms / P)),
                  "sub_batch": sub, "sub_batches": n_sub, "slots": slots, "primed_ticks": prime, "interval_ms": interval_ms,
                  "compute_streams": compute_streams,
                  "path": "resident in HBM (wmx_rt_step_resident)" if resident else
                          "pinned host rows: H2D -> chain -> D2H per sub-batch, overlapped (wmx_rt_submit / _poll / _wait)",
                  "bytes_over_pcie_per_tick": 0 if resident else 2 * rt0.row_bytes * S,
                  "stream_frames_per_s_sustained": S * (interval_ms // 10) / (tick_ms * 1e-3), "failed_steps": failed})
    if gpu_ms:
        out_d["device_ms_between_events_p50"] = round(float(np.median(gpu_ms)), 4)
    if parity:
        T = prime + ticks
        worst, n_off = 0, 0
        for col, s in enumerate(sample):
            want = paced_replay(kind, far, rows, int(pat_of[s]), T, interval_ms)[T - keep:]
            d = np.abs(kept[:, col].astype(np.int32) - want.astype(np.int32))
            worst, n_off = max(worst, int(d.max())), n_off + int((d > 0).sum())
        out_d["parity_checked"] = {"streams": len(sample), "ticks_compared": keep, "ticks_replayed": T, "max_lsb": worst, "samples_off": n_off,
                                   "oracle": "oracle/orc_*.c chain (port), one run per sampled stream over every tick of the run"}
    out_d["wall_s"] = round(time.perf_counter() - t_start, 2)
    return out_d


def paced_main(args, dev):
    """bench.py --paced: one run, or (--paced-search S1,S2,...) the largest S without a miss."""
    from wmix_amd import _lib
    common = dict(sub=args.sub_batch, slots=args.slots, prime=args.paced_prime, resident=args.resident, interval_ms=args.paced_interval_ms or None,
                  compute_streams=args.compute_streams, phases=args.phases, calls=args.calls)
    if args.paced_search:
        runs, s_max = [], None
        for S in sorted(int(x) for x in args.paced_search.split(",")):
            r = run_paced(dev, args.paced_kind, S, args.tick_ms, args.ticks, **common)
            runs.append(r)
            sys.stderr.write("paced S=%d: p50 %.3f p99 %.3f max %.3f ms, %d misses\n" % (S, r["p50_ms"], r["p99_ms"], r["max_ms"], r["misses"]))
            if r["misses"] == 0:
                s_max = S
            else:
                break
        out = {"metric": "largest S without a deadline miss", "value": s_max, "unit": "concurrent streams", "tick_ms": args.tick_ms,
               "ticks_per_run": args.ticks, "runs": runs}
    else:
        r = run_paced(dev, args.paced_kind, args.streams or 65536, args.tick_ms, args.ticks, **common)
        out = {"metric": "paced tick latency", "value": r["p99_ms"], "unit": "ms (p99)", "higher_is_better": False, "realtime": r}
    out.update({"n_gpus": 1, "data": "synthetic", "dtype": "f32", "build": _lib.build_info(), "cpu_model": _cpu_model()})
    print(json.dumps(out))
    sys.stdout.flush()



def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch_ranks(n, argv):
    """`python bench.py --gpus N` without a torchrun environment: start N fresh rank processes of this script (one per GPU)
    and relay rank 0's JSON line.  The parent has imported torch but made no HIP call (importing torch does not
    initialise the GPU), and it starts children -- it never replaces itself with another program."""
    import subprocess
    import tempfile
    import threading
    port = _free_port()
    procs, spill = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", WMIX_BENCH_LAUNCHED_BY="bench.py")
        # ranks > 0 print nothing on success; what they do print (a traceback's first half, a library's chatter) is kept in a
        # temporary file and shown when the launch fails
        f = None if r == 0 else tempfile.TemporaryFile()
        spill.append(f)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else f))
    # rank 0's line is read WHILE the ranks run: a line longer than the pipe's buffer (the per-rank parity records of 8 ranks)
    # would otherwise block rank 0 in write() until the time limit (round-3 ADVICE)
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    # Poll every rank: when one dies (bad device index, library missing on that rank) the others would sit in the rendezvous
    # or in a barrier until the collective's own timeout -- end them, and say which rank failed with what code.
    limit = float(os.environ.get("WMIX_BENCH_LAUNCH_TIMEOUT_S", "3000"))
    t_end = time.monotonic() + limit
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad or all(rc is not None for rc in rcs):
            failed = bad or None
            break
        if time.monotonic() > t_end:
            failed = [(-1, "timeout after %.0f s" % limit)]
            break
        time.sleep(0.05)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=30)
    out0 = b"".join(chunks)
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    if failed:
        for r, f in enumerate(spill):
            if f is not None:
                f.seek(0)
                txt = f.read().decode(errors="replace").strip()
                if txt:
                    sys.stderr.write("---- stdout of rank %d ----\n%s\n" % (r, txt[-4000:]))
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %s; the other ranks were stopped\n" % failed)
        bad_codes = [rc for _, rc in failed if isinstance(rc, int)]
        return bad_codes[0] if bad_codes and 0 < bad_codes[0] < 256 else 1
    return 0


# BASELINE.json configs[0], [1], [3], [4] as (workload, streams per GPU): the default line (configs[2]) carries a short run of each
# (label, workload[, overrides]): the other BASELINE configs, and the headline workload once more in the shape the reference itself
# calls the wrappers with -- handles made with WMIX_INTERVAL_MS = 20, one 20 ms heartbeat (two 10 ms packets) per call
# (src/wmixConf.h:112, src/wmix.c:613-709) -- next to the headline's one packet per launch
SIDE_CONFIGS = [("configs[0]", "g711"), ("configs[1]", "ns"), ("configs[3]", "ns_aec_8k"), ("configs[4]", "ns_agc_mix_32k"),
                ("configs[2] at the daemon's cadence: 20 ms handles, one 20 ms heartbeat per launch", "chain",
                 {"packets_per_step": 2, "interval_ms": 20})]


def _make_workload(cls, dev, n_mine, rank, dist, args, lo=0):
    if issubclass(cls, ChainWorkload):
        return cls(dev, n_mine, rank, dist, args.packets_per_step, args.interval_ms, args.cohorts, args.cohort_layout, args.coalesce,
                   getattr(args, "far_ends", 1), getattr(args, "far_chunk", 1))
    if issubclass(cls, StubCpuWorkload):
        return cls(dev, n_mine, rank, dist, args.packets_per_step, lo, getattr(args, "far_chunk", 1))
    if issubclass(cls, AecmWorkload):
        return cls(dev, n_mine, rank, dist, args.packets_per_step)
    return cls(dev, n_mine, rank)


def _side_config(label, name, args, dev, overrides=None):
    """One of the other BASELINE configs, measured like the headline (same --steps / --warmup / --prime / --spinup, HIP events
    around the timed steps, the dominant kernel bracketed by its own events) and proven in the same run: parity_check replays
    sampled streams through the oracle for exactly the packets fed.  One GPU, no CPU baseline (the workload's own line has it)."""
    import copy
    cls, n = WORKLOADS[name]
    a = copy.copy(args)
    a.packets_per_step, a.interval_ms, a.cohorts, a.cohort_layout, a.coalesce, a.far_ends, a.far_chunk = 1, 10, 1, "arrival", False, 1, 1
    for k, v in (overrides or {}).items():
        setattr(a, k, v)
    t_start = time.perf_counter()
    wl = _make_workload(cls, dev, n, 0, None, a)

    def sync():
        torch.cuda.synchronize()
    n_prime, elapsed, _ = _measure(wl, a, True, sync)
    for _ in range(min(a.steps, 16)):
        wl.step("all")
    sync()
    dom_ms = wl.dominant_ms()
    frac = None
    if dom_ms:
        frac = wl.dominant_bytes_per_frame * wl.n_frames / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    parity = wl.parity_check()
    entry = {"config": label, "workload": wl.name, "streams": n, "value": wl.n_frames * a.steps / elapsed, "unit": "frames/s",
             "ms_per_step": elapsed / a.steps * 1e3, "frames_per_step": wl.n_frames, "steps": a.steps, "warmup": a.warmup, "primed_steps": n_prime, "dtype": wl.dtype,
             "roofline": {"kernel": wl.dominant_kernel, "frac": round(frac, 5) if frac else None,
                          "avg_launch_ms": round(dom_ms, 5) if dom_ms else None,
                          "algorithmic_bytes_per_launch": wl.dominant_bytes_per_frame * wl.n_frames},
             "whole_step_hbm_frac": round(wl.n_frames * a.steps / elapsed * wl.bytes_per_frame / 1e9 / HBM_PEAK_GBS, 5),
             "parity_checked": parity, "wall_s": None}
    for attr in ("chain", "ns", "agc", "mix", "nsx", "aecm"):  # the batch handles give their HBM back before the next config
        h = getattr(wl, attr, None)
        if h is not None and hasattr(h, "close"):
            h.close()
    del wl
    torch.cuda.empty_cache()
    entry["wall_s"] = round(time.perf_counter() - t_start, 2)
    return entry


def _measure(wl, args, on_gpu, sync_all):
    """Priming, warm-up, spin-up and the K timed steps of one workload on this rank.  Returns (primed steps, seconds of the
    timed region on the device, seconds on the host's clock)."""
    n_prime = args.prime + (wl.min_prime() if hasattr(wl, "min_prime") else 0)
    for _ in range(n_prime + args.warmup):
        wl.step(False)
    sync_all()
    # The launch loop is Python: a generation-2 pass of its garbage collector stops the host for ~36 ms (seen at a fixed
    # step of the loop, tools_dev/chain_steps.py) while the GPU runs dry -- 3 % of a 1 000-step region, none of it the
    # measured work.  Collect now, keep the collector off for the timed steps.
    import gc
    gc.collect()
    gc.disable()
    # The barrier + synchronize above (and the one below) bracket the region as the contract asks, but they also leave the
    # device empty and clocked down: a short region started cold measures the ramp, not the path (round 2: 20 steps read 13 %
    # slower than 1 000).  So `--spinup` untimed steps are queued first, WITHOUT a synchronisation behind them, and the K timed
    # steps are bracketed by two HIP events recorded in the launch stream: ms_per_step is the device time between them --
    # exactly K steps, on a device that is already busy.  The host's wall clock over the same K steps is reported beside it.
    on_events = on_gpu
    for _ in range(args.spinup if on_gpu else 0):
        wl.step(False)
    if hasattr(wl, "timed_region"):
        wl.timed_region(True)
    if on_events:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step(True)
    if on_events:
        ev1.record()
    sync_all()
    host_elapsed = time.perf_counter() - t0
    gc.enable()
    elapsed = ev0.elapsed_time(ev1) * 1e-3 if on_events else host_elapsed
    if hasattr(wl, "timed_region"):
        wl.timed_region(False)
    return n_prime, elapsed, host_elapsed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed steps (default 1000: a good second of GPU time for the chain)")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0, help="streams (frames per step) per GPU; 0 = workload default")
    ap.add_argument("--total-streams", type=int, default=0,
                    help="shard THIS many streams over the ranks instead of --streams per GPU: contiguous ranges, the remainder on the "
                         "first ranks (wmix_amd.shard.stream_range); the line then says scaling: strong")
    ap.add_argument("--prime", type=int, default=256,
                    help="untimed steps run before the warm-up so that every stream is past the reference's start-up phases "
                         "(NS: 200 blocks of noise-model start-up, ns_core.c:1103-1160; AEC: pass-through until the far-end "
                         "buffer has filled, echo_cancellation.c:651-657); the timed steps then measure the steady state")
    ap.add_argument("--packets-per-step", type=int, default=1, choices=[1, 2, 4, 8],
                    help="chain workload: 10 ms packets per stream per step / launch (default 1; the daemon itself hands the "
                         "chain 20 ms = 2 packets per call at 16 kHz, src/wmix.c:613-709)")
    ap.add_argument("--interval-ms", type=int, default=10, choices=[10, 20],
                    help="chain workloads: the interval the handles are created with (aec_init / agc_init / vad_init).  20 is the "
                         "daemon's own WMIX_INTERVAL_MS (src/wmixConf.h:112): 20 ms VAD packets, 20 ms AEC packets at 8 kHz; needs an "
                         "even --packets-per-step (the daemon's heartbeat is 2)")
    ap.add_argument("--cohorts", type=int, default=1,
                    help="chain workloads: the streams are N groups of handles created at N distinct ticks (group j joins in front of "
                         "step j: a control plane and a far-end history of its own, wmx_chain_add_cohort); the priming grows to N + "
                         "--prime steps")
    ap.add_argument("--far-ends", type=int, default=1,
                    help="chain workloads: N DISTINCT far-end signals per GPU, stream s cancelled against far-end s * N // S (every mix "
                         "group / call its own far-end: aec_process2's far-end is per handle); the roofline entry then counts the far-end "
                         "spectra each group of S / N streams shares")
    ap.add_argument("--far-chunk", type=int, default=1,
                    help="N > 1 ranks: rank 0 broadcasts the far-end of K steps with ONE collective (SURVEY section 5: one ncclBroadcast per batch "
                         "of K frames; the daemon's far-end is 400 ms old when the AEC gets it, src/wmix.c:651-657: it is known long before)")
    ap.add_argument("--coalesce", action="store_true",
                    help="with --cohorts: wmx_chain_coalesce behind every step (cohorts whose control planes have converged are merged; "
                         "the priming grows until they have)")
    ap.add_argument("--cohort-layout", default="arrival", choices=["arrival", "interleaved"],
                    help="which streams join together: neighbours (slots handed out in arrival order) or stream s in group s %% N")
    ap.add_argument("--spinup", type=int, default=64,
                    help="untimed steps queued directly in front of the timed ones, with no synchronisation in between: the timed "
                         "region starts on a busy device at its working clock (a 20-step region then reads like a 1000-step one)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="default workload on one GPU: leave out the `configs` block (short runs of BASELINE configs[0], [1], [3], [4], "
                         "each with its own in-run parity proof)")
    ap.add_argument("--paced", action="store_true",
                    help="real-time operation instead of back-to-back steps: a host timer releases one tick every --tick-ms; a tick = the "
                         "packages of S streams from pinned host memory through wmx_rt_tick (H2D, NS -> AEC -> AGC -> VAD, D2H); reports "
                         "p50 / p99 / p99.9 / max latency and the misses against the reference's budget, tick - 2 ms (src/wmix.c:536-538)")
    ap.add_argument("--tick-ms", type=float, default=20.0, help="--paced: the tick (WMIX_INTERVAL_MS, src/wmixConf.h:112: 20)")
    ap.add_argument("--ticks", type=int, default=1500, help="--paced: paced ticks per run")
    ap.add_argument("--paced-kind", default="pcm16k", choices=sorted(PACED_KINDS),
                    help="--paced: 16 kHz (or 8 kHz) mono PCM packages of one tick, or the 8 kHz RTP/PCMA packet edge (172-byte datagrams)")
    ap.add_argument("--paced-interval-ms", type=int, default=0, help="--paced: the package a tick carries (default: the tick itself)")
    ap.add_argument("--sub-batch", type=int, default=32768, help="--paced: streams per sub-batch (one wmx_pipe each; uploads and downloads of "
                                                                "neighbouring sub-batches run beside the compute)")
    ap.add_argument("--compute-streams", type=int, default=1, help="--paced: wmx_rt_set_compute_streams (sub-batch b on stream b %% n)")
    ap.add_argument("--calls", action="store_true",
                    help="--paced: every stream hears a far-end of its own (wmx_rt_create_pcm_calls: aec_process2's far-end is per handle; the "
                         "far-end of a call is the other party) -- far rows beside the near rows, 122 KB of far-end history per stream")
    ap.add_argument("--phases", type=int, default=1,
                    help="--paced: release the streams in P groups tick / P apart instead of all at the same instant (every group still has "
                         "the whole tick as its period and tick - 2 ms as its budget)")
    ap.add_argument("--slots", type=int, default=4, help="--paced: sets of pinned rows (tick t works on slot t %% slots)")
    ap.add_argument("--paced-prime", type=int, default=150, help="--paced: unpaced ticks in front (past every stage's start-up)")
    ap.add_argument("--resident", action="store_true", help="--paced: rows resident in HBM, no PCIe inside the tick")
    ap.add_argument("--paced-search", default="", help="--paced: comma-separated stream counts, ascending; stops at the first with a miss")
    ap.add_argument("--no-realtime", action="store_true", help="default workload on one GPU: leave out the short paced run (`realtime` block)")
    args = ap.parse_args()

    if args.paced:
        torch.cuda.set_device(0)
        from wmix_amd import _lib as _l
        _l.lib()
        return paced_main(args, torch.device("cuda", 0))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(_launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks\n" % (args.gpus, world))
        sys.exit(2)

    cls, default_streams = WORKLOADS[args.workload]
    on_gpu = getattr(cls, "needs_gpu", True)
    backend = None
    # WMIX_BENCH_FORCE_DIST=1 (developer check, tests/test_multirank_gpu.py): take the N > 1 code path -- process group,
    # far-end broadcast, gathers, barriers -- with the ranks the launcher started, even if that is one.  On a 1-GPU box this
    # is the only way to put backend nccl (= RCCL) itself under those calls: two ranks cannot share a device under RCCL.
    force_dist = os.environ.get("WMIX_BENCH_FORCE_DIST") == "1" and "WORLD_SIZE" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # developer check of the N > 1 code path on a 1-GPU box: every rank on cuda:0, gloo instead of RCCL
        one_gpu = os.environ.get("WMIX_BENCH_ONE_GPU_GLOO") == "1"
        if one_gpu:
            local_rank = 0
        if on_gpu:
            torch.cuda.set_device(local_rank)
        if one_gpu or not on_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        backend = dist.get_backend()
    else:
        dist = None
        if on_gpu:
            torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if dist is not None else 0) if on_gpu else torch.device("cpu")

    if on_gpu:
        from wmix_amd import _lib
        _lib.lib()  # no fallback: raises when the HIP library is missing

    n_mine, lo = args.streams or default_streams, 0
    if args.total_streams:
        from wmix_amd.shard import stream_range
        lo, hi = stream_range(args.total_streams, rank, world)
        n_mine = hi - lo
        if n_mine < 1:
            sys.stderr.write("bench.py: --total-streams %d leaves rank %d of %d without a stream\n" % (args.total_streams, rank, world))
            sys.exit(2)
    # one rank per GPU: the rank's device IS its LOCAL_RANK, and the collective library sees every rank (checked, not assumed:
    # the first run on a real 8-GPU node must not silently put two ranks on one device)
    rank_device = None
    if on_gpu:
        rank_device = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(),
                       "name": torch.cuda.get_device_name(torch.cuda.current_device()),
                       "uuid": str(getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "uuid", ""))}
        if dist is not None and backend == "nccl":
            assert torch.cuda.current_device() == local_rank, "rank %d: current device %d, LOCAL_RANK %d" % (rank, torch.cuda.current_device(), local_rank)
            assert dist.get_world_size() == world, "RCCL sees %d ranks, the launcher started %d" % (dist.get_world_size(), world)
    wl = _make_workload(cls, dev, n_mine, rank, dist, args, lo)

    def sync_all():
        if on_gpu:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    n_prime, elapsed, host_elapsed = _measure(wl, args, on_gpu, sync_all)
    per_rank_ms = [elapsed / args.steps * 1e3]
    per_rank_frames = [wl.n_frames]
    rank_devices = [rank_device]
    if dist is not None:
        # max over ranks is the job's time; every rank's own figure rides along for the record
        t = torch.tensor([elapsed, float(wl.n_frames)], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank_ms = [float(x[0].item()) / args.steps * 1e3 for x in every]
        per_rank_frames = [int(x[1].item()) for x in every]
        elapsed = max(float(x[0].item()) for x in every)
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, rank_device)
        if backend == "nccl" and world > 1 and os.environ.get("WMIX_BENCH_ONE_GPU_GLOO") != "1":
            devs = [d["device"] for d in rank_devices]
            assert len(set(devs)) == world, "ranks share a device: %s" % rank_devices

    # per-stage breakdown: a few extra steps with every launch bracketed by events, outside the timed region
    for _ in range(min(args.steps, 16)):
        wl.step("all")
    sync_all()

    if hasattr(wl, "measure_pcie") and rank == 0 and (dist is None or not isinstance(wl, ChainWorkload)):
        # (several ranks: the chain workloads stream nothing, and every rank replays its own streams further down)
        parity_early = wl.parity_check()  # before the streaming steps advance the state past what was recorded
        # at least 200 streamed steps whatever --steps says: the pipeline's fill and drain (three slots) are not part of its rate
        wl.measure_pcie(max(min(args.steps, 300), 200))
    else:
        parity_early = None
    frames_total = sum(per_rank_frames) * args.steps  # every rank's own share (equal without --total-streams)
    value = frames_total / elapsed
    dom_ms = wl.dominant_ms()
    roofline = None
    traffic, traffic_src = _pmc_traffic(wl.dominant_kernel, wl.n_frames, getattr(wl, "pmc_tag", "chain"))
    if dom_ms:
        dom_bytes = wl.dominant_bytes_per_frame
        if getattr(wl, "far_ends", 1) > 1:
            # "non-shared far adds xfBuf 6 240 + xfwBuf 6 240" to a handle's live state (read + written once per frame): the
            # S / N streams of a far-end share that history
            dom_bytes += 2.0 * (6240 + 6240) * wl.far_ends / wl.n_streams
        achieved = dom_bytes * wl.n_frames / (dom_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": wl.dominant_kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "traffic_source": traffic_src,
                    "avg_launch_ms": round(dom_ms, 5),
                    "algorithmic_bytes_per_launch": dom_bytes * wl.n_frames,
                    # what actually bounds the per-stream DSP kernels: vector-ALU issue (see _pmc_issue)
                    "valu_issue": _pmc_issue(wl.dominant_kernel, wl.n_frames, getattr(wl, "pmc_tag", "chain"), dom_ms)}
    if roofline is None and getattr(wl, "name", "") == "rtp_chain_8k_pcma":
        step_ms = elapsed / args.steps * 1e3
        achieved = wl.bytes_per_frame * wl.n_frames / (step_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "whole step (ingest, ns, aec far + near, agc, vad, egress)", "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "traffic_source": traffic_src, "avg_launch_ms": round(step_ms, 5),
                    "algorithmic_bytes_per_launch": wl.bytes_per_frame * wl.n_frames}
    out = {
        "metric": "10 ms frames/s", "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "timed_by": ("HIP events recorded in the launch stream around the K timed steps (max over ranks), %d untimed spin-up steps "
                     "queued in front; barrier + synchronize on both sides of spin-up + region" % args.spinup) if on_gpu else "host clock",
        "host_wall_ms_per_step": host_elapsed / args.steps * 1e3,
        "scaling": "strong" if args.total_streams else "weak", "vs_baseline": None, "dtype": wl.dtype, "data": "synthetic",
        "config": dict(wl.config(), primed_steps=n_prime), "roofline": roofline,
        "stage_ms": wl.stage_ms() if hasattr(wl, "stage_ms") else None,
        "stage_ms_source": "mean over up to 16 extra steps after the timed region, outliers beyond 5x the median dropped (inside the timed region only the dominant kernel carries events)",
        "whole_step_hbm_frac": round(value / world * wl.bytes_per_frame / 1e9 / HBM_PEAK_GBS, 5),
        "total_streams": args.total_streams or None,
        "per_rank_ms_per_step": [round(x, 5) for x in per_rank_ms],
        "per_rank_frames_per_step": per_rank_frames,
        "rank_devices": rank_devices if on_gpu else None,
        # the size the collective library itself reports (backend nccl = RCCL on ROCm); None on one rank
        "rccl_ranks": (dist.get_world_size() if dist is not None and backend == "nccl" else None),
        "dist_backend": backend,
        "launched_by": os.environ.get("WMIX_BENCH_LAUNCHED_BY", "torchrun" if "TORCHELASTIC_RUN_ID" in os.environ else "direct"),
        # which build of the library measured this ("default" = the product; anything else is a developer variant, loaded on purpose)
        "build": (_lib.build_info() if on_gpu else None),
    }
    if hasattr(wl, "parity_check"):
        if dist is not None:
            # every rank replays its own sampled streams through the oracle (ranks > 0 received the far-end only through the
            # broadcast); rank 0 prints all of them
            mine = wl.parity_check()
            every = [None] * world
            dist.all_gather_object(every, mine)
            out["parity_checked"] = every[0]
            out["parity_checked_ranks"] = every
        else:
            out["parity_checked"] = (parity_early or wl.parity_check()) if rank == 0 else None
    if rank == 0:
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = wl.cpu_baseline(args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        plain = (args.packets_per_step, args.interval_ms, args.cohorts, args.coalesce, args.streams, args.total_streams, args.far_ends) == (1, 10, 1, False, 0, 0, 1)
        if world == 1 and dist is None and on_gpu and args.workload == DEFAULT_WORKLOAD and plain and not args.no_configs:
            # the other four BASELINE configs on the same line, each measured and proven in this very run; the headline's
            # handles are released first (the 8 kHz config alone holds 131 072 streams of state)
            wl.chain.close()
            del wl
            torch.cuda.empty_cache()
            out["configs"] = [_side_config(c[0], c[1], args, dev, c[2] if len(c) > 2 else None) for c in SIDE_CONFIGS]
            if not args.no_realtime:
                # the headline workload as the reference runs it: PACED.  65 536 concurrent 16 kHz streams, one 20 ms package per stream
                # every 20 ms from pinned host memory and back (6 s of it; the long runs and S_max: profiles/r06/)
                try:
                    out["realtime"] = run_paced(dev, "pcm16k", 65536, 20.0, 300)
                    out["config"]["realtime"] = {k: out["realtime"][k] for k in ("streams", "tick_ms", "budget_ms", "ticks", "p50_ms", "p99_ms", "max_ms", "misses")}
                except Exception as e:  # the paced run is an extra: it must not take the line down
                    out["realtime"] = {"error": repr(e)}
            # the same entries, cut down to what fits any truncation of the line, inside `config`
            out["config"]["configs"] = [{"config": e["config"], "workload": e["workload"], "streams": e["streams"],
                                         "value": round(e["value"], 1), "ms_per_step": round(e["ms_per_step"], 5),
                                         "kernel": e["roofline"]["kernel"], "frac": e["roofline"]["frac"],
                                         "avg_launch_ms": e["roofline"]["avg_launch_ms"],
                                         "parity_max_lsb": e["parity_checked"].get("max_lsb")} for e in out["configs"]]
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _pmc_traffic(kernel, n_frames, tag="chain"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary (profiles/rNN/chain_hbm_pmc.json,
    made by profiles/tools/profile_chain.sh: separate FETCH_SIZE and WRITE_SIZE passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  PMC counters cannot be read from inside this process, so the
    figure is the one measured on the same command and stream count when the profile was taken; None if the
    profile was taken at another size or is absent."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for path in sorted(glob.glob(os.path.join(here, "profiles", "r*", tag + "_hbm_pmc.json")), reverse=True):
        try:
            d = json.load(open(path))
            if d.get("n_frames_per_launch") != n_frames:
                continue
            base = kernel.split("<")[0]
            tot = 0
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                hit = [v for k, v in d[c].items() if k.split("<")[0] == base]
                tot += hit[0]["bytes_per_dispatch_corrected"]
            return tot, os.path.relpath(path, here)
        except Exception:
            continue
    return None, None


def _pmc_issue(kernel, n_frames, tag, launch_ms):
    """Vector-issue accounting of `kernel` (the per-stream DSP kernels are bound by instruction issue, not by HBM).

    Inputs, all committed under profiles/rNN/ and made on the GPU box by profiles/tools/profile_workload.sh:
      <tag>_sq_pmc.json       SQ_INSTS_VALU / SQ_INSTS_SALU per launch (dynamic counts, rocprofv3 PMC pass of this very command)
      issue_costs.json        SIMD time per wave64 instruction of each class at W resident waves per SIMD, every CU busy, from
                              waves that each run for a fixed time (tools_dev/ubench/issue_cost.hip, second version: the
                              first divided a grid's wall time by its instructions and priced the arbiter's unfairness in)
      <tag>_issue_model.json  the kernel's ISA class histogram priced with that table at its occupancy
                              (tools_dev/issue_model.py)
    Figures:
      lower_bound   every vector instruction at the CHEAPEST measured price of any class at the kernel's occupancy: no
                    launch can issue its instructions faster, whatever their mix -- frac = bound / measured launch <= 1;
      mix_estimate  the same count at the histogram's mean price: the time the SIMDs' vector ALUs are busy.  The histogram is
                    static (start-up and rare-update code included), so this one is an estimate, not a bound;
      scalar_ceiling the dynamic scalar-instruction count at the price of a pure scalar stream (s_nop / s_waitcnt at theirs):
                    what the scalar side would take ALONE; beside vector work most of it overlaps (the table's mix classes).
    What is left of the launch after mix_estimate is time in which no wave of a SIMD had a vector instruction ready: one wave
    issues an instruction of any kind every ~5 cycles at best, dependent ones wait for their operands, and these kernels
    hold only 4-5 waves per SIMD (registers, LDS)."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for path in sorted(glob.glob(os.path.join(here, "profiles", "r*", tag + "_sq_pmc.json")), reverse=True):
        try:
            d = json.load(open(path))
            hbm = json.load(open(path.replace("_sq_pmc.json", "_hbm_pmc.json")))
            if hbm.get("n_frames_per_launch") != n_frames:
                continue
            base = kernel.split("<")[0]
            hit = [v for k, v in d.items() if k.split("<")[0] == base]
            insts = hit[0]["mean"]["SQ_INSTS_VALU"]
            out = {"valu_insts_per_frame": round(insts / n_frames, 1), "source": os.path.relpath(path, here)}
            mpath = path.replace("_sq_pmc.json", "_issue_model.json")
            if os.path.exists(mpath):
                m = json.load(open(mpath))
                n_simd = 1024.0  # 256 CUs x 4 SIMDs
                lb = insts * m["cheapest_valu"]["ns"] / n_simd * 1e-6
                est = insts * m["mean_ns_per_valu"] / n_simd * 1e-6
                out.update({"waves_per_simd": m["waves_per_simd"], "priced_at_waves_per_simd": m.get("priced_at_waves_per_simd", m["waves_per_simd"]),
                            "cheapest_ns_per_valu": m["cheapest_valu"]["ns"],
                            "mean_ns_per_valu": m["mean_ns_per_valu"], "lower_bound_ms": round(lb, 5),
                            "lower_bound_frac": round(lb / launch_ms, 4), "mix_estimate_ms": round(est, 5),
                            "mix_estimate_frac": round(est / launch_ms, 4), "model": os.path.relpath(mpath, here)})
                sc, salu = m.get("scalar"), hit[0]["mean"].get("SQ_INSTS_SALU")
                if sc and salu:
                    # SQ_INSTS_SALU counts s_nop / s_waitcnt too: split the dynamic count like the static one
                    idle = sc["static_nop_waitcnt"] / max(1, sc["static_nop_waitcnt"] + sc["static_alu_branch_smem"])
                    t = salu * (idle * sc["ns_nop"] + (1 - idle) * sc["ns_alu"]) / n_simd * 1e-6
                    out.update({"salu_insts_per_frame": round(salu / n_frames, 1), "scalar_ceiling_ms": round(t, 5),
                                "scalar_ceiling_frac": round(t / launch_ms, 4)})
            return out
        except Exception:
            continue
    return None


if __name__ == "__main__":
    main()
