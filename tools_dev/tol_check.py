"""developer check of a TOLERANCE build of the float AEC (WMIX_AMD_LIB=<variant>): the 3 000-frame parity gate's streams
(tests/test_aec_gpu.py::test_chain_parity_gate_3000_frames) through the whole chain on the GPU against the oracle chain; prints the
largest difference inside the first 56 packets (the start-up pass-through and the blocks behind it, where the REFERENCE itself emits
NaN-derived zeros, DESIGN_HISTORY section 2) and from packet 56 on, per rate.  Run on the GPU box: python tools_dev/tol_check.py [n_streams]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(n_pick=64, stages=15):
    import torch
    from oracle import loader as L
    from test_aec_gpu import gpu_chain
    from wmix_amd import synth
    cuda = torch.device("cuda:0")
    for freq in (16000, 8000):
        pkt = freq // 100
        S, n = 256, 3000
        far = synth.far_end(8001, n, pkt)
        near = synth.near_end(8100, S, n, pkt, far=far)
        got = gpu_chain(cuda, 1, freq, stages, far, near, pkts_per_launch=50)
        pick = np.random.default_rng(8).choice(S, n_pick, replace=False)
        port = L.port()
        want = np.stack([L.run_chain(port, 1, freq, 5, stages, far, near[s], pkt, prefix="orc") for s in pick])
        d = np.abs(got[pick].astype(np.int32) - want.astype(np.int32)).reshape(n_pick, n, pkt)
        early, late = d[:, :56], d[:, 56:]
        bad = np.argwhere(late.max(2) > 1)
        print(json.dumps({"lib": os.environ.get("WMIX_AMD_LIB", "default"), "freq": freq, "stages": stages, "streams": n_pick, "packets": n,
                          "max_lsb_packets_0_55": int(early.max()), "max_lsb_packets_56_on": int(late.max()),
                          "samples_off_from_56_on": int((late > 0).sum()), "samples_total_from_56_on": int(late.size),
                          "stream_packets_beyond_1_lsb_from_56_on": len(bad),
                          "first_such": [[int(pick[a]), int(b) + 56] for a, b in bad[:5]]}))


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
