# developer helper: where StreamingPipe's time goes.  PIPE_TIMED=1: a timeline of ticks 20..31 (H2D, compute, D2H from events,
# host wait / submit time); PIPE_AB=1: StreamingPipe against the instrumented copy; default: 3 and 5 slots.
import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from wmix_amd import pipeline, synth
import numpy as np
dev = torch.device('cuda:0')
S = 65536
chain = pipeline.RtpChain(S, dev)
far = torch.zeros((2, 80), dtype=torch.int16, device=dev)
dg = torch.randint(0, 255, (S, 172), dtype=torch.uint8)
dg[:, 0] = 0x80; dg[:, 1] = 8

def run(slots, h2d, d2h, steps=100, cls=None):
    pipeline.StreamingPipe.SLOTS = slots
    pipe = (cls or pipeline.StreamingPipe)(chain)
    for s in range(slots):
        pipe.h_in[s].copy_(dg)
        pipe.d_in[s].copy_(dg)
    for _ in range(6):
        pipe.submit(far)
    pipe.drain(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.submit(far)
    pipe.drain(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

d_in, d_out = torch.empty((S, 172), dtype=torch.uint8, device=dev), torch.empty((S, 172), dtype=torch.uint8, device=dev)
d_in.copy_(dg)
for _ in range(10): chain.step(d_in, far, d_out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): chain.step(d_in, far, d_out)
torch.cuda.synchronize(); print("resident %.3f ms" % ((time.perf_counter() - t0) * 10))
class TimedPipe(pipeline.StreamingPipe):
    """StreamingPipe with timing events around each copy and each tick's compute."""
    def __init__(self, chain):
        super().__init__(chain)
        self.log = []

    def submit(self, far):
        E = lambda: torch.cuda.Event(enable_timing=True)
        s = self.k % self.SLOTS
        self.k += 1
        main = torch.cuda.current_stream()
        th0 = time.perf_counter()
        if self.ev_out[s] is not None:
            self.ev_out[s].synchronize()
        th1 = time.perf_counter()
        a0, a1, c0, c1, b0, b1 = E(), E(), E(), E(), E(), E()
        with torch.cuda.stream(self.s_in):
            a0.record(self.s_in)
            self.d_in[s].copy_(self.h_in[s], non_blocking=True)
            a1.record(self.s_in)
        main.wait_event(a1)
        c0.record(main)
        self.c.step(self.d_in[s], far, self.d_out[s])
        c1.record(main)
        with torch.cuda.stream(self.s_out):
            self.s_out.wait_event(c1)
            b0.record(self.s_out)
            self.h_out[s].copy_(self.d_out[s], non_blocking=True)
            b1.record(self.s_out)
            self.ev_out[s] = b1
        self.log.append((a0, a1, c0, c1, b0, b1, th1 - th0, time.perf_counter() - th1))
        return s

if os.environ.get("PIPE_AB"):
    for _ in range(2):
        print("StreamingPipe %.3f ms   TimedPipe %.3f ms" % (run(3, True, True), run(3, True, True, cls=TimedPipe)))
    sys.exit(0)
if os.environ.get("PIPE_TIMED"):
    pipeline.StreamingPipe.SLOTS = 3
    pipe = TimedPipe(chain)
    for s in range(3):
        pipe.h_in[s].copy_(dg)
    for _ in range(40):
        pipe.submit(far)
    pipe.drain(); torch.cuda.synchronize()
    ref = pipe.log[20][2]
    for i in range(20, 32):
        a0, a1, c0, c1, b0, b1, hw, hs = pipe.log[i]
        print("tick %d: h2d [%.3f..%.3f] compute [%.3f..%.3f] d2h [%.3f..%.3f]  host wait %.3f submit %.3f ms" % (
            i, ref.elapsed_time(a0), ref.elapsed_time(a1), ref.elapsed_time(c0), ref.elapsed_time(c1), ref.elapsed_time(b0), ref.elapsed_time(b1), hw * 1e3, hs * 1e3))
    sys.exit(0)
if os.environ.get("PIPE_TRACE"):
    print("traced: slots 3, both copies, 30 ticks: %.3f ms" % run(3, True, True, 30))
    sys.exit(0)
for slots in (3, 5, 3, 5):
    print("slots", slots, "%.3f ms" % run(slots, True, True))
