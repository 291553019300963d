"""mfft_sizes.py -- transforms/s and HBM rate of wmx_mfft by kind and size (amplitude output for the forward kinds), 2^26 input
samples per launch, HIP events over 100 launches."""
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import os  # noqa: E402

from wmix_amd import _lib, mfft  # noqa: E402

if os.environ.get("WMX_TOOL_LIB"):
    _lib.LIB_PATH = os.environ["WMX_TOOL_LIB"]  # another build of the library, for A/B

for kind, want in ((0, "a"), (1, "a"), (2, "ri"), (3, "ri")):
    for n in (64, 128, 256, 512, 1024, 2048, 4096):
        batch = (1 << 26) // n
        re = torch.randn(batch, n, device="cuda") * 1000
        im = torch.randn(batch, n, device="cuda") * 1000 if kind in (0, 2) else None
        f = lambda: mfft.transform(kind, re, im, want=want)
        for _ in range(5):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        arrays = (2 if im is not None else 1) + len(want)
        print(json.dumps({"kind": kind, "n": n, "batch": batch, "ms": round(ms, 4), "transforms_per_s": round(batch / ms * 1e3),
                          "TBs": round(arrays * 4 * (1 << 26) / ms / 1e9, 2)}), flush=True)
