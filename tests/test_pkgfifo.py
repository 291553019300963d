"""AEC far-end delay FIFO (src/wmix.c:432-526): oracle restatement against the real playPkgBuff_add/get
(oracle/_ref/ref_mix_driver pkgfifo, only where the reference was built) and the device FIFO against the oracle."""
import ctypes as C

import numpy as np
import pytest

from oracle import loader

PKG, SLOTS = 320, 22


class Fifo(C.Structure):
    _fields_ = [("slots", C.c_void_p), ("n_slots", C.c_int), ("pkg_bytes", C.c_int), ("interval_ms", C.c_int), ("frame_bytes", C.c_int),
                ("count", C.c_int)]


def orc_run(port, pkts, delay, slots=SLOTS):
    store = np.zeros(slots * PKG, np.uint8)
    f = Fifo()
    port.orc_pkgfifo_init(C.byref(f), store.ctypes.data_as(C.c_void_p), slots, PKG, 20, 2)
    out, rcs = [], []
    for p in pkts:
        port.orc_pkgfifo_add(C.byref(f), np.ascontiguousarray(p).ctypes.data_as(C.c_void_p))
        o = np.zeros(PKG, np.uint8)
        rcs.append(port.orc_pkgfifo_get(C.byref(f), o.ctypes.data_as(C.c_void_p), delay))
        out.append(o)
    return np.stack(out), rcs


def packets(n, seed=0):
    return np.random.default_rng(seed).integers(0, 256, size=(n, PKG), dtype=np.uint8)


@pytest.mark.skipif(not loader.have_ref_mix(), reason="real reference not built here")
@pytest.mark.parametrize("delay", [400, 0, 20, 40, 380, 420, 440, 1000])
def test_oracle_against_real_reference(oracle_port, delay):
    pk = packets(70, delay)
    want = np.frombuffer(loader.ref_mix("pkgfifo", delay, stdin=pk.tobytes()), np.uint8).reshape(-1, PKG)
    got, rcs = orc_run(oracle_port, pk, delay)
    assert not any(rcs) and np.array_equal(got, want)


@pytest.mark.parametrize("platform", ["hi3516", "t31"])
def test_oracle_against_the_other_platform_builds(oracle_port, platform):
    """AEC_FIFO_PKG_NUM = AEC_INTERVALMS / WMIX_INTERVAL_MS + 2 (src/wmixConf.h:141) is 37 slots in the hi3516 build (700 ms) and 2
    in the t31 build (0 ms).  The real playPkgBuff_add / _get of each build against the
    restatement with that many slots: the build's own delay, and delays on both sides of what the FIFO can hold."""
    if not loader.have_ref_mix(platform):
        pytest.skip("oracle/_ref/ref_mix_driver_%s not present" % platform)
    aec_ms = loader.PLATFORMS[platform][0]
    slots = aec_ms // 20 + 2
    for delay in sorted({aec_ms, 0, 20, 40, max(aec_ms - 20, 0), aec_ms + 20, aec_ms + 40, 1000}):
        pk = packets(100, delay + 1)
        want = np.frombuffer(loader.ref_mix("pkgfifo", delay, stdin=pk.tobytes(), platform=platform), np.uint8).reshape(-1, PKG)
        got, rcs = orc_run(oracle_port, pk, delay, slots=slots)
        assert not any(rcs) and np.array_equal(got, want), delay
    if aec_ms == 0:
        # what the t31 daemon's canceller is given (src/wmix.c:494-510 with delayms 0: pkgCount = count - count = 0, always slot 0,
        # which playPkgBuff_add refreshes every SECOND tick): the package just played on even ticks, the one before it on odd ticks
        pk = packets(10, 3)
        got, _ = orc_run(oracle_port, pk, 0, slots=slots)
        assert np.array_equal(got, pk[np.arange(10) & ~1])


@pytest.mark.skipif(not loader.have_ref_mix(), reason="real reference not built here")
def test_oracle_fractional_delay_where_defined(oracle_port):
    """delays that are not a multiple of the interval prepend the tail of the slot two before; compare wherever the
    reference stays inside its array (the oracle flags the rest)"""
    pk = packets(70, 5)
    for delay in (410, 30, 395):
        want = np.frombuffer(loader.ref_mix("pkgfifo", delay, stdin=pk.tobytes()), np.uint8).reshape(-1, PKG)
        got, rcs = orc_run(oracle_port, pk, delay)
        ok = np.array(rcs) == 0
        assert ok.any() and np.array_equal(got[ok], want[ok])


@pytest.mark.gpu
@pytest.mark.parametrize("delay", [400, 0, 60, 410])
def test_device_fifo_vs_oracle(cuda, oracle_port, delay):
    import torch
    from wmix_amd._lib import WmxError
    from wmix_amd.pkgfifo import PkgFifo
    S = 300
    base = packets(50 * 3, 9).reshape(50, 3, PKG)
    fifo = PkgFifo(S)
    want = [orc_run(oracle_port, base[:, s], delay) for s in range(3)]
    for k in range(50):
        fifo.add(torch.from_numpy(base[k][np.arange(S) % 3]).to(cuda))
        try:
            got = fifo.get(delay).cpu().numpy()
        except WmxError:
            assert want[0][1][k] == -1
            continue
        assert want[0][1][k] == 0
        for s in (0, 1, 2, 298, 299):
            assert np.array_equal(got[s], want[s % 3][0][k]), (k, s)
    fifo.close()
