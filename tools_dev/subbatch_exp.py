"""subbatch_exp.py [--streams N] [--sub S ...] [--steps K] -- the chain of bench.py's default workload with its N streams
split into S contiguous sub-batches, each with its own ChainBatch handle on its own HIP stream: the sub-batches' launches
are independent, so the device can run one sub-batch's noise suppressor beside another's echo canceller and fill the
ramp-up / drain of every launch.  Prints ms per step (all N streams one packet further) for every S."""
import argparse
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from wmix_amd import synth  # noqa: E402
from wmix_amd.chain import AEC, AGC, NS, VAD, ChainBatch  # noqa: E402


def run(n_streams, S, steps, spin):
    dev = torch.device("cuda", 0)
    K, pkt = 200, 160
    far = synth.far_end(3000, K, pkt)
    base = synth.near_end(3001, 256, K, pkt, far=far).reshape(256, K, pkt)
    b = torch.from_numpy(np.ascontiguousarray(base.transpose(1, 0, 2))).to(dev)
    inp = b[:, torch.arange(n_streams, device=dev) % 256]
    far_src = torch.from_numpy(far.reshape(K, pkt).copy()).to(dev)
    per = n_streams // S
    subs = [ChainBatch(per, 1, 16000, 10, 5, NS | AEC | AGC | VAD) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    work2 = [torch.empty_like(inp[0:1]) for _ in range(2)]  # two output buffers, alternating (a forked tail is joined two ticks later)
    main = torch.cuda.current_stream()

    def step(k):
        for i, (c, st) in enumerate(zip(subs, streams)):
            with torch.cuda.stream(st):
                rc, _, _ = c.process_packet_major(far_src[k % K:k % K + 1], inp[k % K:k % K + 1, i * per:(i + 1) * per],
                                                  out=work2[k & 1][:, i * per:(i + 1) * per])
                assert rc == 0

    k = 0
    for _ in range(256 + spin):
        step(k)
        k += 1
    torch.cuda.synchronize()
    for _ in range(spin):
        step(k)
        k += 1
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for st in streams:
        main.wait_stream(st)
    e0.record(main)
    for st in streams:
        st.wait_stream(main)
    for _ in range(steps):
        step(k)
        k += 1
    for st in streams:
        main.wait_stream(st)
    e1.record(main)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    return {"sub_batches": S, "streams": n_streams, "ms_per_step": round(ms, 5), "frames_per_s": round(n_streams / ms * 1e3)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=65536)
    ap.add_argument("--sub", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--spinup", type=int, default=64)
    a = ap.parse_args()
    for S in a.sub:
        print(json.dumps(run(a.streams, S, a.steps, a.spinup)), flush=True)
