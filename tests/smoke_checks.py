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
