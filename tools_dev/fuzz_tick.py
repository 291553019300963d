"""fuzz_tick.py [--cases N] [--seed S] -- random daemons through wmx_tick_* against one oracle daemon per group (oracle.loader.tick_port,
which is pinned on the real functions composed, tests/test_tick_oracle.py): source format and count, record streams per group, which
stages are on, WR_NS_PA, the platform build (echo delay, play-head lead), rwTest switched on for a random span, sources that fall
silent.  Exit code 1 on any mismatch."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
from oracle import loader as L  # noqa: E402
from test_tick_gpu import gpu_tick  # noqa: E402
from test_tick_oracle import tick_inputs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    dev = torch.device("cuda:0")
    port = L.port()
    failed = 0
    for i in range(a.cases):
        platform = str(rng.choice(["alsa", "alsa", "hi3516", "t31"]))
        aec_ms, correct = L.PLATFORMS[platform]
        c = {"case": i, "platform": platform, "src_freq": int(rng.choice([8000, 11025, 16000, 22050, 32000, 44100, 48000])),
             "src_chn": int(rng.integers(1, 3)), "n_src": int(rng.integers(1, 7)), "R": int(rng.integers(1, 4)), "G": int(rng.integers(1, 5)),
             "T": int(rng.integers(60, 150)), "stages": int(rng.integers(1, 16)), "ns_pa": bool(rng.random() < 0.25),
             "rw": bool(rng.random() < 0.3), "silent_from": int(rng.integers(20, 200))}
        per_group = [tick_inputs(9000 + 31 * i + g, c["T"], c["n_src"], c["R"], c["src_freq"], c["src_chn"]) for g in range(c["G"])]
        src = np.stack([p[0] for p in per_group])
        local = np.stack([p[1] for p in per_group])
        src[:, c["silent_from"]:] = 0
        rw = (int(rng.integers(0, 30)), c["T"]) if c["rw"] else None  # on from some tick to the end (the oracle's switch is whole-run:
        if rw:                                                        #  compare from a run that has it on from tick 0)
            rw = (0, c["T"])
        try:
            got = gpu_tick(dev, src, local, c["src_freq"], c["src_chn"], c["stages"] | (64 if c["ns_pa"] else 0), platform=platform, rw_test=rw)
            bad = 0
            for g in range(c["G"]):
                want = L.tick_port(port, src[g], local[g], c["src_freq"], c["src_chn"],
                                   stages=c["stages"] | (16 if c["ns_pa"] else 0) | (32 if rw else 0), aec_delay_ms=aec_ms, play_correct=correct)
                for k in ("play", "far", "out"):
                    bad += int((got[k][g] != want[k]).sum())
            c["samples_differing"] = bad
        except Exception as e:
            c["error"] = repr(e)[:300]
            bad = 1
        if bad:
            failed += 1
            print(json.dumps(c), flush=True)
    print(json.dumps({"cases": a.cases, "failed": failed, "seed": a.seed}))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
