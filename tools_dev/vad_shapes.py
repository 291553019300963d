"""wmx_vad_process per launch at 65 536 streams for every shape vad_init accepts, the four-wave pipeline against the one-lane
kernel (WMIX_AMD_VAD_ONE_LANE=1): one JSON line per shape.  Run on the GPU box: python tools_dev/vad_shapes.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
from make_vadagc_golden import vad_input  # noqa: E402


def timed(chn, freq, ims, one_lane, S=65536, n=60):
    from wmix_amd.vad import VadBatch
    if one_lane:
        os.environ["WMIX_AMD_VAD_ONE_LANE"] = "1"
    else:
        os.environ.pop("WMIX_AMD_VAD_ONE_LANE", None)
    vb = VadBatch(S, chn, freq, ims)
    base = np.stack([vad_input(chn, freq, ims, 1, n_calls=n, seed=40 + s) for s in range(64)]).reshape(64, n, vb.pkt)
    d = torch.from_numpy(np.ascontiguousarray(base.transpose(1, 0, 2))).cuda()[:, torch.arange(S, device="cuda") % 64].contiguous()
    for k in range(10):
        vb.process_packet_major(d[k:k + 1], 1)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n - 10)]
    for k, (a, b) in zip(range(10, n), ev):
        a.record()
        vb.process_packet_major(d[k:k + 1], 1)
        b.record()
    torch.cuda.synchronize()
    vb.close()
    t = np.array([a.elapsed_time(b) for a, b in ev])
    return float(np.median(t)) * 1e3


for chn, freq, ims in ((1, 8000, 10), (1, 16000, 10), (2, 16000, 10), (1, 32000, 10), (2, 32000, 10), (1, 16000, 20), (2, 16000, 20), (2, 8000, 20)):
    pipe, lane = timed(chn, freq, ims, False), timed(chn, freq, ims, True)
    print(json.dumps({"chn": chn, "freq": freq, "interval_ms": ims, "streams": 65536, "pipeline_us": round(pipe, 1), "one_lane_us": round(lane, 1),
                      "us_per_10ms_frame_per_1k_streams": round(pipe / 65.536 / (ims // 10), 3)}), flush=True)
