"""wmx_pipe_submit / wmx_pipe_wait fail clean (round-5 VERDICT weak 6 / next 4).  A variant of the library built with
-DWMX_FAULT_INJECTION (tools_dev/build/lib_faults.so, made by __graft_entry__.build(); refused by the Python mirror like every variant
unless asked for) lets a test make "HIP call n of this entry point" fail.  For every n from 1 to the last call of a submit:

  * the submit returns an error, the rotation / slots in flight / pending download are what they were (the download still owed for the
    step before arrives and is right);
  * a fault BEFORE the first launch has advanced nothing: failed_steps stays, the same rows are submitted again;
  * a fault AFTER it has lost the step (failed_steps counts it): the host restores the streams from the blobs it took at its last
    checkpoint (wmx_chain_export_stream / _export_cohort) and submits again;
  * either way the next three submits deliver exactly what an undisturbed run delivers.

The child process below does the work (it must load the variant library; this process has the product)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

VARIANT = os.path.join(ROOT, "tools_dev", "build", "lib_faults.so")


def _child_rt():
    """the paced tick (wmx_rt_submit over three sub-batches) under the same faults: a failed tick takes its slot, loses the steps of the
    sub-batches it had reached, leaves every pipe's bookkeeping consistent -- and after the host has restored the streams from its
    checkpoint, the following ticks are those of an undisturbed run"""
    import torch
    from wmix_amd import _lib, synth
    from wmix_amd.lifetime import Lifetime
    from wmix_amd.realtime import RtBatch
    L = _lib.lib()
    assert "WMX_FAULT_INJECTION" in _lib.build_info()
    L.wmx_debug_fail_nth_hip_call.argtypes = [C.c_long]
    L.wmx_debug_hip_calls.restype = C.c_long
    dev = torch.device("cuda:0")
    S, n, slots = 24, 300, 3
    far = synth.far_end(9720, n, 160).reshape(n, 1, 160)
    near = synth.near_end(9721, S, n, 160, far=far.reshape(-1)).reshape(S, n, 160).transpose(1, 0, 2)
    make = lambda: RtBatch(S, dev, sub_batch=10, slots=slots, kind="pcm", chn=1, freq=16000, interval_ms=10)  # noqa: E731

    class Borrowed(Lifetime):
        _mod = "chain"

        def __init__(self, h, n_streams):
            self._h, self.n_streams = h, n_streams

    u = make()
    want = np.zeros((n, S, 160), np.int16)
    for k in range(n):
        u.fill(k % slots, near[k])
        u.h_far[k % slots][:] = far[k]
        assert u.tick(None) == k % slots
        want[k] = u.gather(k % slots)
    u.close()

    f = make()
    chains = [Borrowed(L.wmx_pipe_chain(p), nb) for p, nb in zip(f.pipes, f.batch_n)]
    nxt = [0]

    def tick(k, arm=0):
        slot = nxt[0]
        nxt[0] = (slot + 1) % slots  # a tick takes its slot whatever becomes of it
        f.fill(slot, near[k])
        f.h_far[slot][:] = far[k]
        got = C.c_int(-1)
        if arm:
            L.wmx_debug_fail_nth_hip_call(arm)
        rc = L.wmx_rt_submit(f._h, None, C.byref(got), None)
        calls = L.wmx_debug_hip_calls()
        L.wmx_debug_fail_nth_hip_call(0)
        if got.value == -1:  # the call failed in front of everything (its device scope): not even the slot was taken
            assert rc != 0
            nxt[0] = slot
        else:
            assert got.value == slot
        rw = L.wmx_rt_wait(f._h)
        return rc, rw, slot, calls
    k = 0
    for _ in range(2 * slots):
        rc, rw, slot, _ = tick(k)
        assert rc == 0 and rw == 0 and np.array_equal(f.gather(slot), want[k])
        k += 1
    report = {"kind": "rt", "faults": 0, "lost_steps": 0}
    i = 1
    while True:
        assert k + 3 < n
        snap = [([c.export_stream(s) for s in range(c.n_streams)], c.export_cohort(0)) for c in chains]
        lost0 = f.failed_steps()
        rc, rw, slot, calls = tick(k, arm=i)
        if rc == 0:
            assert calls < i and np.array_equal(f.gather(slot), want[k])
            k += 1
            break
        assert rc <= -11000 and rw == 0, (i, rc, rw)
        report["faults"] += 1
        report["lost_steps"] += f.failed_steps() - lost0
        for c, (streams, cohort) in zip(chains, snap):  # the checkpoint: whatever each sub-batch had reached is undone
            c.import_cohort(0, cohort)
            for s in range(c.n_streams):
                c.import_stream(s, streams[s], cohort=0)
        for kk in (k, k + 1, k + 2):
            rc, rw, slot, _ = tick(kk)
            assert rc == 0 and rw == 0, (i, kk, L.wmx_last_error())
            assert np.array_equal(f.gather(slot), want[kk]), ("tick %d after fault %d" % (kk, i))
        k += 3
        i += 3 if i > 12 else 1  # every call of the first sub-batch's uploads and launches, then every third
    f.close()
    print(json.dumps(report))


def _child(kind):
    if kind == "rt":
        return _child_rt()
    import torch
    from wmix_amd import _lib, synth
    from wmix_amd.lifetime import Lifetime
    from wmix_amd.pipeline import PcmChain, RtpChain, StreamingPipe
    L = _lib.lib()
    assert "WMX_FAULT_INJECTION" in _lib.build_info()
    L.wmx_debug_fail_nth_hip_call.argtypes = [C.c_long]
    L.wmx_debug_hip_calls.restype = C.c_long
    dev = torch.device("cuda:0")
    S, n = 24, 400

    class Borrowed(Lifetime):  # the pipe's own chain, for the checkpoint blobs
        _mod = "chain"

        def __init__(self, h):
            self._h, self.n_streams = h, S

    if kind == "pcm":
        far = synth.far_end(9700, n, 160)
        near = synth.near_end(9701, S, n, 160, far=far).reshape(S, n, 160).transpose(1, 0, 2)
        make = lambda: PcmChain(S, dev, 1, 16000, 10, 5, 15, slots=3)  # noqa: E731
        far_rows = far.reshape(n, 1, 160)
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle import loader
        from test_pipeline_gpu import make_datagrams
        far, pk = make_datagrams(loader.port(), S, n, seed=9710)
        near = pk.transpose(1, 0, 2)
        make = lambda: RtpChain(S, dev)  # noqa: E731
        far_rows = far.reshape(n, 2, 80)

    def run(pipe, k):
        slot = k % pipe.SLOTS
        pipe.h_in[slot][:] = near[k]
        pipe.h_far[slot][:] = far_rows[k]
        s = C.c_int(-1)
        rc = L.wmx_pipe_submit(pipe.c._h, None, C.byref(s), None)
        return rc, s.value

    # the undisturbed run
    u = make()
    pu = StreamingPipe(u)
    want = np.zeros((n,) + pu.h_out[0].shape, pu.h_out[0].dtype)
    for k in range(n):
        rc, slot = run(pu, k)
        assert rc == 0
        pu.wait(slot)
        want[k] = pu.h_out[slot]
    u.close()

    f = make()
    pf = StreamingPipe(f)
    chain = Borrowed(L.wmx_pipe_chain(f._h))
    got = {}
    k = 0
    for _ in range(6):  # past the first rotation: every submit from here on makes the same runtime calls
        rc, slot = run(pf, k)
        assert rc == 0
        k += 1
    pf.wait(-1)
    report = {"kind": kind, "faults": 0, "before_first_launch": 0, "lost_steps": 0, "wait_faults": 0}
    i = 1
    while True:
        assert k + 4 < n, "ran out of steps at fault %d" % i
        # step k - 1 stays owed (submitted, its download not queued yet) while step k fails
        rc, slot_prev = run(pf, k)
        assert rc == 0
        k_prev, k = k, k + 1
        snap = None
        if kind == "pcm":  # the host's checkpoint: every stream and the cohort
            snap = ([chain.export_stream(s) for s in range(S)], chain.export_cohort(0))
        lost0 = L.wmx_pipe_failed_steps(f._h)
        L.wmx_debug_fail_nth_hip_call(i)
        rc, _ = run(pf, k)
        calls = L.wmx_debug_hip_calls()
        L.wmx_debug_fail_nth_hip_call(0)
        if rc == 0:  # the fault lies behind the last call of a submit: every call has had its turn
            assert calls < i
            pf.wait(-1)
            k += 1  # (that submit was step k, and it went through)
            break
        assert rc <= -11000, (i, rc)  # a HIP error's code, far from the reference's -1
        report["faults"] += 1
        lost = L.wmx_pipe_failed_steps(f._h) - lost0
        assert lost in (0, 1)
        if lost == 0:
            report["before_first_launch"] += 1
        else:
            report["lost_steps"] += 1
            if kind != "pcm":
                # the RTP senders' sequence numbers have no import call: this form is walked through the faults in front of the first launch only
                break
            chain.import_cohort(0, snap[1])
            for s in range(S):
                chain.import_stream(s, snap[0][s], cohort=0)
        # the same step again, then two more; the step that was owed when the fault came is delivered by the next submit / wait
        for kk in (k, k + 1, k + 2):
            rc, slot = run(pf, kk)
            assert rc == 0, (i, kk, L.wmx_last_error())
            if kk == k:
                pf.wait(slot_prev)
                assert np.array_equal(pf.h_out[slot_prev], want[k_prev]), ("the step owed at fault %d" % i)
            pf.wait(slot)
            assert np.array_equal(pf.h_out[slot], want[kk]), ("step %d after fault %d" % (kk, i))
        k += 3
        i += 1
    # wmx_pipe_wait: a failing wait changes nothing either
    for j in (1, 2, 3, 4):
        rc, slot = run(pf, k)
        assert rc == 0
        L.wmx_debug_fail_nth_hip_call(j)
        rw = L.wmx_pipe_wait(f._h, slot)
        L.wmx_debug_fail_nth_hip_call(0)
        report["wait_faults"] += rw != 0
        pf.wait(slot)
        assert np.array_equal(pf.h_out[slot], want[k]), ("wait fault", j)
        k += 1
    f.close()
    print(json.dumps(report))


@pytest.mark.parametrize("kind", ["pcm", "rtp", "rt"])
def test_every_fallible_call_of_a_submit(cuda, kind):
    # __graft_entry__.build() makes it; `make` brings it up to date when a source has changed since (nothing to do otherwise, a minute of
    # hipcc if the tree was not built that way at all)
    subprocess.run([os.path.join(ROOT, "tools_dev", "variant.sh"), "build", "faults", "-DWMX_FAULT_INJECTION"], timeout=1500, capture_output=True)
    assert os.path.exists(VARIANT), "tools_dev/build/lib_faults.so is built by __graft_entry__.build() (tools_dev/variant.sh build faults -DWMX_FAULT_INJECTION)"
    env = dict(os.environ, WMIX_AMD_LIB=VARIANT, WMIX_AMD_ALLOW_VARIANT_BUILD="1", PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), kind], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    if kind == "rt":
        assert rep["faults"] >= 20 and rep["lost_steps"] >= 10, rep
        return
    assert rep["before_first_launch"] >= 3 and rep["wait_faults"] >= 2, rep
    if kind == "pcm":
        assert rep["lost_steps"] >= 10 and rep["faults"] == rep["before_first_launch"] + rep["lost_steps"], rep


def test_the_product_build_has_no_fault_hooks(wmx):
    assert not hasattr(wmx, "wmx_debug_fail_nth_hip_call") and not hasattr(wmx, "wmx_debug_hip_calls")


if __name__ == "__main__":
    _child(sys.argv[1])
