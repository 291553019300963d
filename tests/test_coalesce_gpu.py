"""Coalescing control cohorts (wmx_aec_coalesce / wmx_chain_coalesce): handles of the reference that are created at different
ticks but report the same delay end up with control planes that differ only in where their rings stand (W:modules/audio_processing/
aec/echo_cancellation.c:599-872, aec_core.c:1719-1850 are index arithmetic on fill levels) -- a batch that gained one cohort per
join tick folds them into a handful, and every stream still produces what its own handle of the reference would have produced
from its own first packet: far-end history, far power, the re-blocking rings' phase, and the comfort-noise generator
(randomization_functions.c:94-112: one state per handle, advanced by 64 draws per block of THAT handle)."""
import numpy as np
import pytest
import torch

from oracle import loader as L
from test_aec_gpu import check_float_path
from wmix_amd import synth
from wmix_amd.aec import AecBatch
from wmix_amd.chain import ChainBatch

pytestmark = pytest.mark.gpu


def _inputs(cuda, seed, S, K, U, pkt):
    far = synth.far_end(seed, K, pkt).reshape(K, pkt)
    base = synth.near_end(seed + 1, U, K, pkt, far=far.reshape(-1)).reshape(U, K, pkt)
    inp = torch.from_numpy(np.ascontiguousarray(base.transpose(1, 0, 2))).to(cuda)[:, torch.arange(S, device=cuda) % U]  # [K, S, pkt]
    return far, base, inp, torch.from_numpy(far.copy()).to(cuda)


@pytest.mark.parametrize("freq,S,T", [(16000, 4096, 1500), (8000, 1024, 1500)])
def test_cohorts_of_random_join_ticks_fold_into_a_handful(cuda, oracle_port, freq, S, T):
    K, U = 200, 64
    pkt = freq // 100
    far, base, inp, dfar = _inputs(cuda, 8100 + freq // 1000, S, K, U, pkt)
    rng = np.random.default_rng(5)
    n_ticks = 150
    ticks = np.concatenate([[0], np.sort(rng.choice(np.arange(1, 800), n_ticks - 1, replace=False))])
    join = ticks[rng.integers(0, n_ticks, S)]
    for i, t in enumerate(ticks):  # every join tick has a stream
        join[i] = t
    by_tick = {int(t): np.flatnonzero(join == t).astype(np.int32) for t in ticks}
    pick = sorted({0, int(np.argmax(join))} | {int(x) for x in rng.choice(S, 22, replace=False)})
    dpick = torch.tensor(pick, device=cuda)
    rec = torch.empty(T, len(pick), pkt, dtype=torch.int16, device=cuda)
    work = torch.empty_like(inp[0:1])
    cb = ChainBatch(S, 1, freq, 10, 5, n_cohorts=1)
    active = np.zeros(S, np.uint8)
    merged, peak = [], 0
    for t in range(T):
        if t in by_tick:
            c = 0 if t == 0 else cb.add_cohort()
            if t == 0:
                cb.reset_cohort(0)
            cb.reset_streams(by_tick[t], cohort=c)
            active[by_tick[t]] = 1
            cb.set_active(active)
        peak = max(peak, cb.live_cohorts())
        rc, codes, _ = cb.process_packet_major(dfar[t % K:t % K + 1], inp[t % K:t % K + 1], out=work)
        assert rc == 0 and not codes.any()
        rec[t] = work[0, dpick]
        for fr, to in cb.coalesce(32):
            merged.append((t, fr, to))
    live, ids = cb.live_cohorts(), cb.n_cohorts
    cb.close()
    # dozens of handles' control planes ran side by side before the first could fold (a cohort's far-end rings must have filled: 250
    # blocks); at the end there is one per phase of the 64-in-160 (64-in-80) re-blocking -- 2 at 16 kHz, 4 at 8 kHz -- and the id range
    # has shrunk behind them.  (The core's noise-floor and delay-estimate counters count with the stream, not with the cohort.)
    classes = 2 if freq == 16000 else 4
    assert peak >= 20 and live <= classes and ids <= 16 and len(merged) >= n_ticks - classes, (peak, live, ids, len(merged))
    got = rec.cpu().numpy()
    far_seq = np.concatenate([far[t % K] for t in range(T)])
    for col, s in enumerate(pick):
        a = int(join[s])
        near = np.concatenate([base[s % U, t % K] for t in range(a, T)])
        want = L.run_chain(oracle_port, 1, freq, 5, 15, far_seq[a * pkt:], near, pkt, prefix="orc").reshape(T - a, pkt)
        check_float_path(got[a:, col], want, max_fraction=1e-4)


def test_only_cohorts_that_report_the_same_delay_fold(cuda, oracle_port):
    """Three delay classes -- 0 ms, 40 ms, and one cohort whose reported delay wanders -- and one cohort that is switched off for 60
    ticks early in its life: nothing folds across the classes, the wanderer stays alone, the sleeper folds only once its rings agree
    with another cohort's again, and every stream matches its own handle."""
    S, T, freq, pkt = 48, 1400, 16000, 160
    far = synth.far_end(8200, T, pkt).reshape(T, pkt)
    near = synth.near_end(8201, S, T, pkt, far=far.reshape(-1)).reshape(S, T, pkt)
    ab = AecBatch(S, 1, freq, 10, n_cohorts=1)
    d = torch.from_numpy(np.ascontiguousarray(near.transpose(1, 0, 2))).to(cuda)
    dfar = torch.from_numpy(far.copy()).to(cuda)
    rng = np.random.default_rng(11)
    n_co = 12
    start = [0] + sorted(int(x) for x in rng.choice(np.arange(1, 300), n_co - 1, replace=False))
    delay_of = [0, 40, 0, 40, 0, 40, 0, 0, 40, 40, 0, 0]  # by creation order
    WANDER, SLEEPER = 6, 7
    members = {k: np.arange(4 * k, 4 * k + 4, dtype=np.int32) for k in range(n_co)}
    cid = {}        # creation order -> current cohort id
    delays_used = {k: [] for k in range(n_co)}
    on_used = {k: [] for k in range(n_co)}
    active = np.zeros(S, np.uint8)
    folds = []
    for t in range(T):
        for k in range(n_co):
            if start[k] == t:
                cid[k] = 0 if k == 0 else ab.add_cohort()
                if k == 0:
                    ab.reset_cohort(0)
                ab.reset_streams(members[k], cohort=cid[k])
                active[members[k]] = 1
                ab.set_active(active)
        G = ab.n_cohorts
        dl, on = np.zeros(G, np.int32), np.zeros(G, np.uint8)
        for k, c in cid.items():
            if c < 0:
                continue
            v = delay_of[k] if k != WANDER else int(20 + 15 * np.sin(t / 37.0) + (t % 7))
            o = 0 if (k == SLEEPER and start[k] + 40 <= t < start[k] + 100) else 1
            dl[c], on[c] = v, o
        for k, c in cid.items():  # what each handle of the reference was called with (folded cohorts: the cohort they joined)
            cc = c
            while cc < 0:
                cc = cid[-cc - 1]
            delays_used[k].append(int(dl[cc]))
            on_used[k].append(int(on[cc]))
        rc, codes = ab.run_cohorts(dfar[t:t + 1], d[t:t + 1].transpose(0, 1), dl, cohort_on=on)
        assert rc == 0 and not codes.any()
        for fr, to in ab.coalesce(8):
            kf = [k for k, c in cid.items() if c == fr][0]
            kt = [k for k, c in cid.items() if c == to][0]
            folds.append((t, kf, kt))
            cid[kf] = -kt - 1  # follows kt from here on
    out = d.cpu().numpy().transpose(1, 0, 2)
    ab.close()
    for t, kf, kt in folds:
        assert delay_of[kf] == delay_of[kt] and WANDER not in (kf, kt), (t, kf, kt)
        # not while it sleeps, nor before its far-end rings have filled again behind the 60 packets it missed
        assert not (SLEEPER in (kf, kt) and t < start[SLEEPER] + 100 + 90), (t, kf, kt)
    assert len(folds) >= 4, folds
    port = oracle_port
    for k in range(n_co):
        a = start[k]
        s = int(members[k][1])
        if k == SLEEPER:  # the handle is simply not called for 60 ticks: far-end and near-end of those ticks never reach it
            keep = np.array(on_used[k], bool)
            tt = np.arange(a, T)[keep]
            want = L.run_aec_delays(port, 1, freq, 10, far[tt].reshape(-1), near[s, tt].reshape(-1), pkt, np.array(delays_used[k])[keep], prefix="orc")
            check_float_path(out[s, tt].reshape(-1), want, max_fraction=1e-4)
            continue
        want = L.run_aec_delays(port, 1, freq, 10, far[a:].reshape(-1), near[s, a:].reshape(-1), pkt, np.array(delays_used[k]), prefix="orc")
        check_float_path(out[s, a:].reshape(-1), want, max_fraction=1e-4)


def test_a_stream_of_a_folded_cohort_migrates_with_its_own_generator(cuda, oracle_port):
    """Export / import after a fold: the blob carries the stream's block count -- the state of its comfort-noise generator -- and the
    batch it moves to makes its noise table reach the oldest stream it was given."""
    S, T, freq, pkt = 8, 1500, 16000, 160
    far = synth.far_end(8300, T, pkt).reshape(T, pkt)
    near = synth.near_end(8301, S, T, pkt, far=far.reshape(-1)).reshape(S, T, pkt)
    d = torch.from_numpy(np.ascontiguousarray(near.transpose(1, 0, 2))).to(cuda)
    dfar = torch.from_numpy(far.copy()).to(cuda)
    ab = AecBatch(S, 1, freq, 10, n_cohorts=1)
    active = np.zeros(S, np.uint8)
    start = {0: 0, 1: 8, 2: 16, 3: 24}  # cohort k: streams 2k, 2k + 1; 8 ticks apart = the same phase
    folds = 0
    for t in range(1200):
        for k, t0 in start.items():
            if t == t0:
                c = 0 if k == 0 else ab.add_cohort()
                if k == 0:
                    ab.reset_cohort(0)
                ab.reset_streams(np.array([2 * k, 2 * k + 1], np.int32), cohort=c)
                active[[2 * k, 2 * k + 1]] = 1
                ab.set_active(active)
        rc, _ = ab.run_cohorts(dfar[t:t + 1], d[t:t + 1].transpose(0, 1), np.zeros(ab.n_cohorts, np.int32))
        assert rc == 0
        folds += len(ab.coalesce(8))
    assert folds == 3 and ab.live_cohorts() == 1 and ab.n_cohorts == 1
    blobs = [ab.export_stream(s) for s in range(S)]
    cblob = ab.export_cohort(0)
    ab.close()
    nb = AecBatch(S, 1, freq, 10, n_cohorts=1)
    nb.import_cohort(0, cblob)
    for s in range(S):
        nb.import_stream(s, blobs[s], cohort=0)
    for t in range(1200, T):
        rc, _ = nb.run_cohorts(dfar[t:t + 1], d[t:t + 1].transpose(0, 1), np.zeros(1, np.int32))
        assert rc == 0
    nb.close()
    out = d.cpu().numpy().transpose(1, 0, 2)
    for k, t0 in start.items():
        for s in (2 * k, 2 * k + 1):
            want = L.run_aec(oracle_port, 1, freq, 10, far[t0:].reshape(-1), near[s, t0:].reshape(-1), pkt, 0, prefix="orc")
            check_float_path(out[s, t0:].reshape(-1), want, max_fraction=1e-4)


def test_a_stream_that_has_lived_for_hours_keeps_its_own_noise_generator(cuda, oracle_port):
    """The comfort-noise generator is part of the STREAM's state (AS_NSEED), as it is of the handle in the reference: a state imported
    with a generator that stands 3.1 million blocks into its life (3.4 hours at 16 kHz) draws what a handle of the reference draws
    whose generator stands there (oracle: the same run with aec->seed set to 777 advanced by 64 draws per block), while its
    neighbours in the batch -- same cohort, same control plane -- draw from their own.  Nothing in the batch grows with a
    stream's age (round-4 ADVICE: the table of rows that did is gone)."""
    from wmix_amd._lib import lib
    S, T, freq, pkt, R = 4, 300, 16000, 160, 3_100_003
    far = synth.far_end(8400, T, pkt).reshape(T, pkt)
    near = synth.near_end(8401, S, T, pkt, far=far.reshape(-1)).reshape(S, T, pkt)
    d = torch.from_numpy(near.copy()).to(cuda)
    dfar = torch.from_numpy(far.copy()).to(cuda)
    ab = AecBatch(S, 1, freq, 10)
    words = lib().wmx_aec_state_words(ab._h)
    blob = ab.export_stream(2)
    hdr = blob.size - 4 * words
    w = blob[hdr:].view(np.uint32)
    assert w[words - 5] == 777  # AS_NSEED = AS_SCAL + 11 of 16 scalar words at the end of the block: aec->seed of a new handle
    w[words - 5] = L.lcg_after_blocks(R)
    ab.import_stream(2, blob)
    rc, _ = ab.process2(dfar, d, delay_ms=0)
    assert rc == 0
    out = d.cpu().numpy()
    ab.close()
    for s in range(S):
        if s == 2:
            want = L.run_aec_seeded(oracle_port, 1, freq, 10, far.reshape(-1), near[s].reshape(-1), pkt, 0, L.lcg_after_blocks(R))
        else:
            want = L.run_aec(oracle_port, 1, freq, 10, far.reshape(-1), near[s].reshape(-1), pkt, 0, prefix="orc")
        check_float_path(out[s].reshape(-1), want, max_fraction=1e-4)
    # and the generator's position is audible in the output: the same stream as a NEW handle gives other samples
    fresh = L.run_aec(oracle_port, 1, freq, 10, far.reshape(-1), near[2].reshape(-1), pkt, 0, prefix="orc")
    assert int((fresh.astype(np.int32) != out[2].reshape(-1)).sum()) > 100


@pytest.mark.parametrize("freq,S,T", [(16000, 2048, 1200), (8000, 1024, 2200)])
def test_aecm_cohorts_fold_too_bit_exact(cuda, oracle_port, freq, S, T):
    """wmx_aecm_coalesce: the fixed-point canceller's control plane has no periodic counters, so cohorts fold as soon as the younger
    one's far-end slab equals the older one's -- the 256-block history is the quick part, the binary far spectrum's running thresholds
    (integer IIRs with a shift of 6) take some 1 400 blocks to meet bit for bit: one cohort per phase of the 80-in-64 re-blocking
    remains (two at 16 kHz, four at 8 kHz).  Integer arithmetic: every sampled stream equals its own handle of the reference bit for bit."""
    from wmix_amd.aecm import AecmBatch
    pkt = freq // 100
    far = synth.far_end(8500 + freq // 1000, T, pkt).reshape(T, pkt)
    U = 32
    base = synth.near_end(8501, U, T, pkt, far=far.reshape(-1)).reshape(U, T, pkt)
    d = torch.from_numpy(np.ascontiguousarray(base.transpose(1, 0, 2))).to(cuda)[:, torch.arange(S, device=cuda) % U].contiguous()  # [T, S, pkt]
    dfar = torch.from_numpy(far.copy()).to(cuda)
    rng = np.random.default_rng(6)
    n_ticks = 60
    ticks = np.concatenate([[0], np.sort(rng.choice(np.arange(1, 250), n_ticks - 1, replace=False))])
    join = ticks[rng.integers(0, n_ticks, S)]
    for i, t in enumerate(ticks):
        join[i] = t
    by_tick = {int(t): np.flatnonzero(join == t).astype(np.int32) for t in ticks}
    pick = sorted({0, int(np.argmax(join))} | {int(x) for x in rng.choice(S, 14, replace=False)})
    ab = AecmBatch(S, 1, freq, 10, n_cohorts=1)
    active = np.zeros(S, np.uint8)
    folds, peak = 0, 0
    for t in range(T):
        if t in by_tick:
            c = 0 if t == 0 else ab.add_cohort()
            if t == 0:
                ab.reset_cohort(0)
            ab.reset_streams(by_tick[t], cohort=c)
            active[by_tick[t]] = 1
            ab.set_active(active)
        peak = max(peak, ab.live_cohorts())
        rc, codes = ab.run_cohorts(dfar[t:t + 1], d[t:t + 1].transpose(0, 1), np.zeros(ab.n_cohorts, np.int32))
        assert rc == 0 and not codes.any()
        folds += len(ab.coalesce(32))
    live, ids = ab.live_cohorts(), ab.n_cohorts
    ab.close()
    classes = 2 if freq == 16000 else 4
    assert peak >= 20 and live <= classes and ids <= 16 and folds >= n_ticks - classes, (peak, live, ids, folds)
    out = d.cpu().numpy()
    for s in pick:
        a = int(join[s])
        want = L.run_aecm(oracle_port, 1, freq, 10, far[a:].reshape(-1), base[s % U, a:].reshape(-1), pkt, 0, prefix="orc")
        assert np.array_equal(out[a:, s].reshape(-1), want), s


def test_coalesce_argument_edges(cuda):
    """Bad arguments are refused; a batch with one cohort, or a call without room to report a merge, merges nothing; cohort_key says
    why a cohort is (not) a candidate."""
    import ctypes as C
    from wmix_amd._lib import lib
    S, freq, pkt = 8, 16000, 160
    ab = AecBatch(S, 1, freq, 10, n_cohorts=1)
    n = C.c_int(7)
    fr, to = np.zeros(4, np.int32), np.zeros(4, np.int32)
    L_ = lib()
    assert L_.wmx_aec_coalesce(None, 1, fr.ctypes.data, to.ctypes.data, 4, C.byref(n), None) != 0
    assert L_.wmx_aec_coalesce(ab._h, -1, fr.ctypes.data, to.ctypes.data, 4, C.byref(n), None) != 0
    assert L_.wmx_aec_coalesce(ab._h, 4, None, None, 4, C.byref(n), None) != 0
    assert L_.wmx_aec_coalesce(ab._h, 4, fr.ctypes.data, to.ctypes.data, 4, C.byref(n), None) == 0 and n.value == 0  # one cohort
    assert ab.cohort_key(0) is None  # still in its start-up phase
    far = synth.far_end(8600, 700, pkt).reshape(700, pkt)
    near = synth.near_end(8601, S, 700, pkt, far=far.reshape(-1)).reshape(S, 700, pkt)
    d = torch.from_numpy(np.ascontiguousarray(near.transpose(1, 0, 2))).to(cuda)
    dfar = torch.from_numpy(far.copy()).to(cuda)
    ab.reset_cohort(0)
    active = np.zeros(S, np.uint8)
    for t in range(700):
        if t in (0, 2):
            c = 0 if t == 0 else ab.add_cohort()
            m = np.arange(0, 4, dtype=np.int32) if t == 0 else np.arange(4, 8, dtype=np.int32)
            ab.reset_streams(m, cohort=c)
            active[m] = 1
            ab.set_active(active)
        rc, _ = ab.run_cohorts(dfar[t:t + 1], d[t:t + 1].transpose(0, 1), np.zeros(ab.n_cohorts, np.int32))
        assert rc == 0
        # no room to report: the pair is proposed and checked again and again, and never merged
        assert L_.wmx_aec_coalesce(ab._h, 4, None, None, 0, C.byref(n), torch.cuda.current_stream().cuda_stream) == 0 and n.value == 0
    assert ab.live_cohorts() == 2
    k0, k1 = ab.cohort_key(0), ab.cohort_key(1)
    assert k0 is not None and np.array_equal(k0, k1)  # two packets apart: the same phase of the re-blocking
    merged = []
    for _ in range(4):
        merged += ab.coalesce(4)
        torch.cuda.synchronize()
    assert merged == [(1, 0)] and ab.live_cohorts() == 1 and ab.n_cohorts == 1
    ab.close()
