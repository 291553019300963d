"""The device's WebRtcSpl_SqrtFloor (spl_fx.h: v_sqrt_f32 of the float-converted argument + two integer corrections)
restated with numpy float32 -- same conversion rounding, same correctly rounded square root -- against the integer square
root, on every argument where the floor changes (k*k - 1, k*k, k*k + 1 for all k < 46 341), the largest arguments, and
10^7 random ones.  W:common_audio/signal_processing/spl_sqrt_floor.c is a 16-step bit loop computing the same floor."""
import math

import numpy as np


def device_sqrt_floor(v):
    v = np.asarray(v, dtype=np.int64)
    out = np.zeros(v.shape, dtype=np.int64)
    pos = v > 0
    u = v[pos].astype(np.uint32)
    r = np.sqrt(u.astype(np.float32)).astype(np.uint32).astype(np.uint64)  # float -> uint32 truncates like v_cvt_u32_f32
    u64 = u.astype(np.uint64)
    r = r - (r * r > u64)
    r = r + ((r + 1) * (r + 1) <= u64)
    out[pos] = r.astype(np.int64)
    return out


def isqrt(v):
    return np.array([math.isqrt(int(x)) if x > 0 else 0 for x in v], dtype=np.int64)


def test_sqrt_floor_boundaries():
    k = np.arange(0, 46341, dtype=np.int64)
    args = np.concatenate([k * k - 1, k * k, k * k + 1, [2**31 - 1, 2**31 - 2, 0, -1, -2**31]])
    args = args[(args >= -2**31) & (args <= 2**31 - 1)]
    assert np.array_equal(device_sqrt_floor(args), isqrt(args))


def test_sqrt_floor_random():
    rng = np.random.default_rng(5)
    args = rng.integers(0, 2**31, size=10_000_000, dtype=np.int64)
    got = device_sqrt_floor(args)
    # r = floor(sqrt(v))  <=>  r*r <= v < (r+1)*(r+1)
    assert np.all(got * got <= args) and np.all((got + 1) * (got + 1) > args)
