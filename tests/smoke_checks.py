"""Body of __graft_entry__.smoke(): one small invocation of each finished stage of the
hot path on the GPU, checked against the oracle (oracle/ is test infrastructure, so the
check lives under tests/)."""
import ctypes as C


def run(dev, np, torch):
    from oracle import loader
    from wmix_amd import g711
    port = loader.port()
    rng = np.random.default_rng(0)
    pcm = rng.integers(-32768, 32768, size=80 * 64, dtype=np.int16)
    for law in "au":
        want = np.zeros(pcm.size, np.uint8)
        getattr(port, "orc_PCM2G711" + law)(C.c_void_p(pcm.ctypes.data), C.c_void_p(want.ctypes.data), pcm.size * 2, 0)
        got = g711.encode(law, torch.from_numpy(pcm).to(dev))
        assert np.array_equal(got.cpu().numpy(), want), "G.711 %s-law encode mismatch" % law
        back = np.zeros(pcm.size, np.int16)
        getattr(port, "orc_G711%s2PCM" % law)(C.c_void_p(want.ctypes.data), C.c_void_p(back.ctypes.data), want.size, 0)
        assert np.array_equal(g711.decode(law, got).cpu().numpy(), back), "G.711 %s-law decode mismatch" % law


def _chain(dev, np, torch):
    """NS -> AEC -> AGC -> VAD on 4 streams x 40 packets of 16 kHz mono vs the oracle chain."""
    from oracle import loader
    from wmix_amd import synth
    from wmix_amd.aec import AecBatch
    from wmix_amd.agc import AgcBatch
    from wmix_amd.ns import NsBatch
    from wmix_amd.vad import VadBatch
    port = loader.port()
    S, n, pkt = 4, 40, 160
    far = synth.far_end(11, n, pkt)
    near = synth.near_end(12, S, n, pkt, far=far)
    want = np.stack([loader.run_chain(port, 1, 16000, 5, 15, far, near[s], pkt, prefix="orc") for s in range(S)])
    d = torch.from_numpy(near.reshape(S, n, pkt).copy()).to(dev)
    dfar = torch.from_numpy(far.reshape(n, pkt).copy()).to(dev)
    ns, aec, agc, vad = NsBatch(S, 1, 16000), AecBatch(S, 1, 16000), AgcBatch(S, 1, 16000, 5), VadBatch(S, 1, 16000)
    for f in range(0, n, 10):
        blk = d[:, f:f + 10]
        ns.process(blk)
        rc, _ = aec.process2(dfar[f:f + 10], blk)
        assert rc == 0
        agc.process(blk)
        vad.process(blk)
    got = d.cpu().numpy().reshape(S, -1)
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff.max() == 0, "chain differs from the oracle by %d LSB" % diff.max()


def _mfft(dev, np, torch):
    """math/fft.c: 8 real 256-point transforms vs the oracle (re / im / amplitude bit-exact)."""
    from oracle import loader
    from wmix_amd import mfft
    port = loader.port()
    x = (np.random.default_rng(3).standard_normal((8, 256)) * 2000).astype(np.float32)
    got = mfft.transform(1, torch.from_numpy(x).to(dev), None, want="ria")
    for b in range(8):
        want = loader.mfft(port, 1, x[b], None, 256, prefix="orc", want="ria")
        for k, w in want.items():
            assert np.array_equal(got[k][b].cpu().numpy().view(np.uint32), w.view(np.uint32)), "math/fft FFTR %s mismatch" % k


def _tick(dev, np, torch):
    """The daemon's tick for 2 mixers x 3 sources x 1 record stream, 60 ticks: mix -> drain -> delay FIFO -> far-end of the group's
    record chain -> zoom, against one oracle daemon per group (played package and far-end bit for bit, record stream too)."""
    from oracle import loader
    from wmix_amd import synth
    from wmix_amd.tick import TickBatch
    port = loader.port()
    G, T, N = 2, 60, 160
    ins = [synth.conference_inputs(40 + g, T, 3, 1, 8000, 1) for g in range(G)]
    tb = TickBatch(G, 1)
    prev = torch.zeros((G, N), dtype=torch.int16, device=dev)
    got = {"play": [], "far": [], "out": []}
    for t in range(T):
        src = torch.zeros((G, 3, N + 1), dtype=torch.int16, device=dev)
        src[:, :, :N] = torch.from_numpy(np.stack([ins[g][0][t] for g in range(G)])).to(dev)
        tb.load(src, 2 * N, 8000, 1)
        play = torch.zeros((G, N), dtype=torch.int16, device=dev)
        far = tb.play(play).clone()
        line = torch.cat([prev, far], 1).to(torch.int32)
        local = torch.from_numpy(np.stack([ins[g][1][t, 0] for g in range(G)])).to(dev).to(torch.int32)
        rec = torch.clamp(local + (line[:, N - loader.TICK_ECHO_DELAY: 2 * N - loader.TICK_ECHO_DELAY] >> 1), -32768, 32767).to(torch.int16)
        tb.record(rec)
        prev = far
        for k, v in (("play", play), ("far", far), ("out", rec)):
            got[k].append(v.cpu().numpy())
    tb.close()
    for g in range(G):
        want = loader.tick_port(port, ins[g][0], ins[g][1], 8000, 1)
        assert np.array_equal(np.stack([x[g] for x in got["play"]]), want["play"]), "tick: played package differs"
        assert np.array_equal(np.stack([x[g] for x in got["far"]]), want["far"]), "tick: far-end out of the delay FIFO differs"
        d = np.abs(np.stack([x[g] for x in got["out"]]).astype(np.int32) - want["out"][:, 0].astype(np.int32))
        assert d.max() == 0, "tick: record stream differs from the oracle daemon by %d LSB" % d.max()


_run_g711 = run


def run(dev, np, torch):  # noqa: F811
    _run_g711(dev, np, torch)
    _chain(dev, np, torch)
    _mfft(dev, np, torch)
    _tick(dev, np, torch)
