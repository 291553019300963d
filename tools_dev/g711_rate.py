"""g711_rate.py -- G.711 encode / decode rate against a device copy of the same number of bytes, by size (HIP events, 200 launches)."""
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from wmix_amd import g711  # noqa: E402


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n in (1 << 24, 1 << 26, 83886080, 1 << 28, 1 << 30):
    pcm = torch.randint(-32768, 32767, (n,), dtype=torch.int16, device="cuda")
    code = torch.empty(n, dtype=torch.uint8, device="cuda")
    back = torch.empty_like(pcm)
    a = torch.empty(3 * n // 2 // 4, dtype=torch.int32, device="cuda")  # 1.5 n bytes in, 1.5 n bytes out = 3 n bytes moved
    b = torch.empty_like(a)
    t_enc = timed(lambda: g711.encode("u", pcm, code))
    t_dec = timed(lambda: g711.decode("u", code, back))
    t_cp = timed(lambda: b.copy_(a))
    gb = 3 * n / 1e9
    print(json.dumps({"samples": n, "MB": round(gb * 1e3), "encode_ms": round(t_enc, 4), "decode_ms": round(t_dec, 4), "copy_ms": round(t_cp, 4),
                      "encode_TBs": round(gb / t_enc, 2), "decode_TBs": round(gb / t_dec, 2), "copy_TBs": round(gb / t_cp, 2)}), flush=True)
