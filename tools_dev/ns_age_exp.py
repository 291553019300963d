"""developer experiment: does ns_kernel's time depend on HOW the streams' ages (blocks since ns_init) are spread over a batch?
65 536 streams, the bench's input; the streams are reset at 256 staggered ticks either in contiguous groups of 256 ("arrival") or
stream s at tick s % 256 ("interleaved"), or all at tick 0; then the kernel is timed per tick with events."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from wmix_amd import synth
from wmix_amd.ns import NsBatch
from wmix_amd.nsx import NsxBatch


def run(layout, S=65536, N=256, K=200, pkt=160, ticks=700, mod="ns"):
    dev = torch.device("cuda:0")
    far = synth.far_end(3000, K, pkt)
    base = synth.near_end(3001, 256, K, pkt, far=far).reshape(256, K, pkt)
    inp = torch.from_numpy(np.ascontiguousarray(base.transpose(1, 0, 2))).to(dev)[:, torch.arange(S, device=dev) % 256]
    work = torch.empty_like(inp[0:1])
    ns = (NsBatch if mod == "ns" else NsxBatch)(S, 1, 16000)
    sidx = np.arange(S)
    join = np.zeros(S, np.int64) if layout == "same" else ((sidx % N) if layout == "interleaved" else (sidx * N // S))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(ticks)]
    for t in range(ticks):
        if layout != "same" and t < N:
            ns.reset_streams(np.flatnonzero(join == t).astype(np.int32))
        ev[t][0].record()
        ns.process_packet_major(inp[t % K:t % K + 1], out=work)
        ev[t][1].record()
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    ns.close()
    return {"module": mod, "layout": layout, "mean_ms_ticks_300_700": float(ms[300:].mean()), "min": float(ms[300:].min()), "max": float(ms[300:].max()),
            "by_100": [round(float(ms[i:i + 100].mean()), 4) for i in range(0, ticks, 100)]}


if __name__ == "__main__":
    mod = sys.argv[1] if len(sys.argv) > 1 else "ns"
    for lay in ("same", "arrival", "interleaved"):
        print(json.dumps(run(lay, mod=mod)))
