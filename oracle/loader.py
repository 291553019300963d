"""oracle/loader.py -- TEST INFRASTRUCTURE ONLY.

ctypes access to (a) oracle/build/liboracle.so, our C restatement of the
reference's hot path, and (b) oracle/_ref/libwmixref.so + ref_mix_driver, the
real reference compiled from /root/reference by oracle/Makefile (present only
where it was prebuilt).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (wmix_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
PORT_SO = os.path.join(HERE, "build", "liboracle.so")
REF_SO = os.path.join(REF_DIR, "libwmixref.so")
REF_MIX = os.path.join(REF_DIR, "ref_mix_driver")

_i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build_port():
    subprocess.check_call(["make", "-s", "-C", HERE, "port"])


def build_ref():
    subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def have_ref():
    return os.path.exists(REF_SO)


def have_ref_mix(platform="alsa"):
    if platform != "alsa":
        exe = ref_mix_exe(platform)
        return os.path.exists(exe) and os.access(exe, os.X_OK)
    return os.path.exists(REF_MIX) and os.access(REF_MIX, os.X_OK)


_port = None
_ref = None
_bound = {}
_bind_lock = __import__("threading").Lock()


def _fn(lib, name, restype, argtypes):
    """The foreign function with its prototype set ONCE: bench.py's all-core CPU baseline calls the run_* helpers from many
    threads, and re-assigning argtypes on the shared function object while another thread converts arguments races."""
    key = (id(lib), name)
    f = _bound.get(key)
    if f is None:
        with _bind_lock:
            f = _bound.get(key)
            if f is None:
                f = getattr(lib, name)
                f.restype = restype
                f.argtypes = argtypes
                _bound[key] = f
    return f


def port():
    """Our C restatement (built on demand: plain gcc, a few seconds).  WMIX_ORACLE_SAN=1: the AddressSanitizer +
    UndefinedBehaviorSanitizer build of the same sources (oracle/Makefile `san`; needs the ASan runtime preloaded)."""
    global _port
    if _port is None:
        if os.environ.get("WMIX_ORACLE_SAN") == "1":
            subprocess.check_call(["make", "-s", "-C", HERE, "san"])
            _port = C.CDLL(os.path.join(HERE, "build", "liboracle_san.so"))
        else:
            build_port()
            _port = C.CDLL(PORT_SO)
    return _port


def ref():
    """The real reference wrappers, pinned to the generic-C AEC kernels."""
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_SO)
        assert lib.ref_pin_generic_c() == 1
        _ref = lib
    return _ref


# ---------------------------------------------------------------- G.711
def _g711(lib, name, src, out_dtype, n_arg):
    fn = getattr(lib, name)
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    src = np.ascontiguousarray(src)
    out = np.zeros(src.size, dtype=out_dtype)
    r = fn(src.ctypes.data, out.ctypes.data, n_arg, 0)
    return out, r


def g711_encode(lib, law, pcm):
    """law 'a'|'u'; pcm int16 -> uint8 codes via PCM2G711x(in,out,bytes,0)."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    return _g711(lib, "PCM2G711" + law, pcm, np.uint8, pcm.size * 2)


def g711_decode(lib, law, codes):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    return _g711(lib, "G711%s2PCM" % law, codes, np.int16, codes.size)


# ---------------------------------------------------------------- wrappers (whole-run drivers)
def run_ns(lib, chn, freq, pcm, frames_per_call, prefix="ref"):
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    out = np.empty_like(pcm)
    n_calls = pcm.size // (frames_per_call * chn)
    fn = _fn(lib, prefix + "_run_ns", C.c_int, [C.c_int, C.c_int, _i16p, _i16p, C.c_int, C.c_int])
    rc = fn(chn, freq, pcm, out, frames_per_call, n_calls)
    assert rc == 0, rc
    return out


def run_nsx(lib, chn, freq, pcm, frames_per_call, prefix="ref"):
    """The wrapper built with MAKE_WEBRTC_NSX (src/webrtc.c:512-521): fixed-point noise suppressor."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    out = np.empty_like(pcm)
    n_calls = pcm.size // (frames_per_call * chn)
    fn = _fn(lib, prefix + "_run_nsx", C.c_int, [C.c_int, C.c_int, _i16p, _i16p, C.c_int, C.c_int])
    rc = fn(chn, freq, pcm, out, frames_per_call, n_calls)
    assert rc == 0, rc
    return out


def run_agc(lib, chn, freq, value, pcm, frames_per_call, prefix="ref"):
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    out = np.empty_like(pcm)
    n_calls = pcm.size // (frames_per_call * chn)
    fn = _fn(lib, prefix + "_run_agc", C.c_int, [C.c_int, C.c_int, C.c_int, _i16p, _i16p, C.c_int, C.c_int])
    rc = fn(chn, freq, value, pcm, out, frames_per_call, n_calls)
    assert rc == 0, rc
    return out


def run_agc_handle(lib, chn, freq, value, pcm, frames_per_call, additions=None, prefix="ref"):
    """One AGC HANDLE driven call by call: agc_init(chn, freq, 10, value), then agc_process per call of frames_per_call frames,
    with agc_addition(fp, v) in FRONT of call c for every (c, v) of `additions` (src/webrtc.c:694-753, 767-839).  prefix "ref":
    the real wrapper functions of oracle/_ref/libwmixref.so; "orc": the restatement's handle API."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    out = np.empty_like(pcm)
    per = frames_per_call * chn
    n_calls = pcm.size // per
    add = dict(additions or {})
    vp = C.c_void_p
    if prefix == "orc":
        init = _fn(lib, "orc_agc_init", vp, [C.c_int, C.c_int, C.c_int, C.c_int])
        run = _fn(lib, "orc_agc_run", C.c_int, [vp, vp, vp, C.c_int])
        addf = _fn(lib, "orc_agc_addition", None, [vp, C.c_uint8])
        rel = _fn(lib, "orc_agc_release", None, [vp])
        h = init(chn, freq, 10, value)
    else:
        init = _fn(lib, "agc_init", vp, [C.c_int, C.c_int, C.c_int, C.c_int, vp])
        run = _fn(lib, "agc_process", C.c_int, [vp, vp, vp, C.c_int])
        addf = _fn(lib, "agc_addition", None, [vp, C.c_uint8])
        rel = _fn(lib, "agc_release", None, [vp])
        h = init(chn, freq, 10, value, None)
    assert h, "agc_init returned NULL"
    try:
        for c in range(n_calls):
            if c in add:
                addf(h, int(add[c]))
            assert run(h, pcm[c * per:].ctypes.data, out[c * per:].ctypes.data, frames_per_call) == 0
    finally:
        rel(h)
    return out


def run_vad(lib, chn, freq, interval_ms, pcm, frames_per_call, prefix="ref"):
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    out = np.empty_like(pcm)
    n_calls = pcm.size // (frames_per_call * chn)
    fn = _fn(lib, prefix + "_run_vad", C.c_int, [C.c_int, C.c_int, C.c_int, _i16p, _i16p, C.c_int, C.c_int])
    rc = fn(chn, freq, interval_ms, pcm, out, frames_per_call, n_calls)
    assert rc == 0, rc
    return out


def run_aec(lib, chn, freq, interval_ms, far, near, frames_per_call, delay_ms=0, prefix="ref"):
    far = np.ascontiguousarray(far, dtype=np.int16)
    near = np.ascontiguousarray(near, dtype=np.int16)
    out = np.empty_like(near)
    n_calls = near.size // (frames_per_call * chn)
    fn = _fn(lib, prefix + "_run_aec", C.c_int, [C.c_int, C.c_int, C.c_int, _i16p, _i16p, _i16p, C.c_int, C.c_int, C.c_int])
    rc = fn(chn, freq, interval_ms, far, near, out, frames_per_call, n_calls, delay_ms)
    assert rc == 0, rc
    return out


def run_aec_seeded(lib, chn, freq, interval_ms, far, near, frames_per_call, delay_ms, seed):
    """orc_run_aec for a handle whose comfort-noise generator stands at `seed` (port only)."""
    far = np.ascontiguousarray(far, dtype=np.int16)
    near = np.ascontiguousarray(near, dtype=np.int16)
    out = np.empty_like(near)
    n_calls = near.size // (frames_per_call * chn)
    fn = _fn(lib, "orc_run_aec_seeded", C.c_int, [C.c_int, C.c_int, C.c_int, _i16p, _i16p, _i16p, C.c_int, C.c_int, C.c_int, C.c_uint32])
    rc = fn(chn, freq, interval_ms, far, near, out, frames_per_call, n_calls, delay_ms, int(seed))
    assert rc == 0, rc
    return out


def lcg_after_blocks(n_blocks):
    """The AEC's comfort-noise generator (seed 777, x -> 69069 x + 1 mod 2^31, 64 draws per block) after n_blocks blocks."""
    a, c, k, x = 69069, 1, 64 * int(n_blocks), 777
    m = (1 << 32) - 1
    while k:
        if k & 1:
            x = (a * x + c) & m
        c = (a * c + c) & m
        a = (a * a) & m
        k >>= 1
    return x & 0x7FFFFFFF


def run_aec_delays(lib, chn, freq, interval_ms, far, near, frames_per_call, delays, prefix="ref"):
    """aec_process2 per call with the reported delay of that call (delays: one per call)."""
    far = np.ascontiguousarray(far, dtype=np.int16)
    near = np.ascontiguousarray(near, dtype=np.int16)
    out = np.empty_like(near)
    n_calls = near.size // (frames_per_call * chn)
    d = np.ascontiguousarray(delays, dtype=np.int32)
    assert d.shape == (n_calls,)
    fn = _fn(lib, prefix + "_run_aec_delays", C.c_int,
             [C.c_int, C.c_int, C.c_int, _i16p, _i16p, _i16p, C.c_int, C.c_int, np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")])
    rc = fn(chn, freq, interval_ms, far, near, out, frames_per_call, n_calls, d)
    assert rc == 0, rc
    return out


def run_aecm(lib, chn, freq, interval_ms, far, near, frames_per_call, delay_ms=0, split=0, prefix="ref", expect_rc=0):
    """The wrapper built with the reference's AECM switch (src/webrtc.c:168-191): fixed-point echo canceller.
    split=1 drives aec_setFrameFar + aec_process instead of aec_process2."""
    far = np.ascontiguousarray(far, dtype=np.int16)
    near = np.ascontiguousarray(near, dtype=np.int16)
    out = np.zeros_like(near)
    n_calls = near.size // (frames_per_call * chn)
    fn = _fn(lib, prefix + "_run_aecm", C.c_int, [C.c_int, C.c_int, C.c_int, _i16p, _i16p, _i16p, C.c_int, C.c_int, C.c_int, C.c_int])
    rc = fn(chn, freq, interval_ms, far, near, out, frames_per_call, n_calls, delay_ms, split)
    assert rc == expect_rc, rc
    return out


def run_chain(lib, chn, freq, agc_value, stages, far, near, frames_per_call, prefix="ref", interval_ms=10):
    """stages bitmask: 1 NS, 2 AEC, 4 AGC, 8 VAD (daemon order, src/wmix.c:613-709); interval_ms: what aec_init / agc_init /
    vad_init are given (the daemon: WMIX_INTERVAL_MS = 20, src/wmixConf.h:112)."""
    far = np.ascontiguousarray(far, dtype=np.int16)
    near = np.ascontiguousarray(near, dtype=np.int16)
    out = np.empty_like(near)
    n_calls = near.size // (frames_per_call * chn)
    fn = _fn(lib, prefix + "_run_chain_iv", C.c_int,
             [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, _i16p, _i16p, _i16p, C.c_int, C.c_int])
    rc = fn(chn, freq, interval_ms, agc_value, stages, far, near, out, frames_per_call, n_calls)
    assert rc == 0, rc
    return out


def run_rtp_chain(lib, far, datagrams, agc_value=5):
    """One stream through the packet edge + chain of SURVEY 8f-1 with the restatement: datagrams uint8 [n, 172] (RTP/PCMA,
    20 ms at 8 kHz) -> orc_rtp_ingest -> NS -> AEC -> AGC -> VAD (160-sample calls, 10 ms packets) -> orc_rtp_egress with a
    fresh sender.  far int16 [n * 160].  Returns uint8 [n, 172]."""
    n = datagrams.shape[0]
    ing = _fn(lib, "orc_rtp_ingest", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p])
    pcm = np.zeros(n * 160, np.int16)
    for k in range(n):
        p = np.ascontiguousarray(datagrams[k])
        assert ing(p.ctypes.data, pcm[k * 160:].ctypes.data, None) == 320
    out = run_chain(lib, 1, 8000, agc_value, 15, far, pcm, 160, prefix="orc")
    snd = (C.c_uint8 * 16)()
    _fn(lib, "orc_rtp_sender_init", None, [C.c_void_p, C.c_int])(snd, 0)
    eg = _fn(lib, "orc_rtp_egress", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p])
    res = np.zeros((n, 172), np.uint8)
    for k in range(n):
        assert eg(snd, 1, 8000, out[k * 160:].ctypes.data, 320, 1, 8000, res[k].ctypes.data) == 172
    return res


# ---------------------------------------------------------------- math/fft.c
MFFT_KINDS = ("FFT", "FFTR", "IFFT", "IFFTR")


def mfft(lib, kind, re, im, n, prefix="ref", want="riap"):
    """One transform of math/fft.c.  kind 0..3 = FFT, FFTR, IFFT, IFFTR; `re` / `im` float32[n] or None (read as
    zeros, like the reference's NULL); `want` picks the outputs (r, i, a = amplitude, p = phase; the inverses have
    no a / p).  Returns a dict of float32[n] arrays."""
    f32 = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
    re, im = f32(re), f32(im)
    ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    out = {k: np.zeros(n, np.float32) for k in want if not (kind >= 2 and k in "ap")}
    o = lambda k: ptr(out.get(k))
    if prefix == "orc":
        fn = lib.orc_mfft
        fn.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_uint]
        fn.restype = C.c_int
        assert fn(kind, ptr(re), ptr(im), o("r"), o("i"), o("a"), o("p"), n) == 0
    else:
        fn = getattr(lib, MFFT_KINDS[kind])
        fn.restype = None
        if kind < 2:
            fn.argtypes = [C.c_void_p] * 6 + [C.c_uint]
            fn(ptr(re), ptr(im), o("r"), o("i"), o("a"), o("p"), n)
        else:
            fn.argtypes = [C.c_void_p] * 4 + [C.c_uint]
            fn(ptr(re), ptr(im), o("r"), o("i"), n)
    return out


def mfft_stream(lib, chunks, st_len, prefix="ref"):
    """fft_stream over successive chunks; returns (stream, [af per call], [pf per call])."""
    stream = np.zeros(st_len, np.float32)
    fn = lib.orc_mfft_stream if prefix == "orc" else lib.fft_stream
    fn.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p]
    fn.restype = C.c_int if prefix == "orc" else None
    afs, pfs = [], []
    for ch in chunks:
        ch = np.ascontiguousarray(ch, np.float32)
        af, pf = np.zeros(st_len, np.float32), np.zeros(st_len, np.float32)
        fn(ch.ctypes.data, len(ch), stream.ctypes.data, st_len, af.ctypes.data, pf.ctypes.data)
        afs.append(af)
        pfs.append(pf)
    return stream, afs, pfs


# ---------------------------------------------------------------- resample + mix (restatement, in-process)
class MixRing(C.Structure):  # orc_mix_ring (oracle/orc_mix.h)
    _fields_ = [("chn", C.c_int), ("freq", C.c_int), ("size", C.c_uint32), ("buff", C.c_void_p), ("head_off", C.c_uint32),
                ("tick", C.c_uint32), ("reduce_mode", C.c_uint8), ("play_correct", C.c_uint32)]


def mix_bind(p):
    for n in ("orc_len_of_out", "orc_len_of_in", "orc_pcm_zoom", "orc_load_data"):
        getattr(p, n).restype = C.c_uint32


def mix_zoom(p, ic, ifr, x, oc, ofr):
    """orc_pcm_zoom of one int16 buffer: the bytes wmix_pcm_zoom would write, as int16."""
    mix_bind(p)
    # room for what the walk can write: one output frame per input frame or ofr / ifr of them, whichever is more (x.size * 16 was
    # too little for 1 x 5000 -> 2 x 48000: tools_dev/fuzz_mix.py found the heap corruption in THIS helper)
    out = np.zeros((int(np.ceil(x.size / ic * max(ofr / ifr, 1.0))) + 8) * oc + 64, np.int16)
    m = p.orc_pcm_zoom(ic, ifr, x.ctypes.data_as(C.c_void_p), x.size * 2, oc, ofr, out.ctypes.data_as(C.c_void_p))
    return out[: m // 2].copy()


def mix_load(p, ring_chn, ring_freq, freq, chn, rmode, rarg, nsrc, sbytes, start, src, play_correct=None):
    """nsrc wmix_load_data calls in order (source i at src + i * sbytes bytes, each with a fresh head) into one fresh ring whose
    head stands at byte `start`.  play_correct: VIEW_PLAY_CORRECT of the platform build in bytes (None = platform/alsa's formula,
    which orc_mix_ring_init applies).  Returns (the ring as int16, [(tick, head) after each call])."""
    mix_bind(p)
    size = ring_chn * 2 * ring_freq
    store = np.zeros(size + 64, np.uint8)
    r = MixRing()
    p.orc_mix_ring_init(C.byref(r), store.ctypes.data_as(C.c_void_p), ring_chn, ring_freq)
    r.head_off, r.reduce_mode = start, rmode
    if play_correct is not None:
        r.play_correct = play_correct
    meta = []
    for i in range(nsrc):
        tick = C.c_uint32(0)
        h = p.orc_load_data(C.byref(r), C.c_void_p(src.ctypes.data + i * sbytes), sbytes, freq, chn, 16, C.c_uint32(0xFFFFFFFF), rarg,
                            C.byref(tick))
        meta.append((tick.value, h))
    return store[:size].view(np.int16).copy(), np.array(meta, np.uint32)


# ---------------------------------------------------------------- the daemon's tick, composed (SURVEY 8f-2 closed: FIFO -> AEC)
TICK_ECHO_DELAY = 40  # samples the room delays the played far-end by on its way into the microphone (the harness' input model)


def tick_room(local, far, prev_far):
    """near = sat(local + (far-end delayed by TICK_ECHO_DELAY samples) >> 1): what a microphone next to the loudspeaker picks up.
    local [.., N] int16 (the room without the loudspeaker), far / prev_far [N]: this tick's and the previous tick's far-end package."""
    line = np.concatenate([prev_far, far]).astype(np.int32)
    n = far.size
    echo = line[n - TICK_ECHO_DELAY: 2 * n - TICK_ECHO_DELAY] >> 1
    return np.clip(local.astype(np.int32) + echo, -32768, 32767).astype(np.int16)


class _PkgFifo(C.Structure):  # orc_pkgfifo (oracle/orc_pkgfifo.c)
    _fields_ = [("slots", C.c_void_p), ("n_slots", C.c_int), ("pkg_bytes", C.c_int), ("interval_ms", C.c_int), ("frame_bytes", C.c_int),
                ("count", C.c_int)]


def tick_port(lib, sources, local, src_freq, src_chn, stages=15, agc_value=5, aec_delay_ms=400, play_correct=None):
    """ONE daemon (1 x 8000 Hz ring, 20 ms packages) over T ticks with the restatement, in the play thread's order (src/wmix.c:
    1347-1440 with wmix_shmem_write_circle inside): the task threads' orc_load_data calls (sources int16 [T, n_src, samples of
    20 ms]; every source keeps its cursor), the drain of one package, orc_pkgfifo add / get(aec_delay_ms) = the far-end, the room
    (tick_room), and per record stream (local int16 [T, n_rec, 160]) NS -> AEC(far) -> AGC -> VAD (`stages` bits 1 2 4 8; bit 16 =
    WR_NS_PA over the playback; bit 32 = wmix->rwTest: record stream 0's output goes back into the play ring through orc_load_data
    with a cursor of its own, src/wmix.c:714-732 -- a feedback loop through loudspeaker, room and cancellers) and the zoom to
    1 x 8000.  aec_delay_ms / play_correct: the platform build's AEC_INTERVALMS / VIEW_PLAY_CORRECT (PLATFORMS; defaults =
    platform/alsa).  Returns dict(play [T,160], far [T,160], near [T,n_rec,160], out [T,n_rec,160], zoom [T,n_rec,160])."""
    mix_bind(lib)
    T, n_src, per = sources.shape
    n_rec, N = local.shape[1], 160
    size = 16000
    store = np.zeros(size + 64, np.uint8)
    ring = store[:size].view(np.int16)
    r = MixRing()
    lib.orc_mix_ring_init(C.byref(r), store.ctypes.data_as(C.c_void_p), 1, 8000)
    r.reduce_mode = 1
    if play_correct is not None:
        r.play_correct = play_correct
    n_slots = aec_delay_ms // 20 + 2
    fstore = np.zeros(n_slots * 2 * N, np.uint8)
    f = _PkgFifo()
    lib.orc_pkgfifo_init(C.byref(f), fstore.ctypes.data_as(C.c_void_p), n_slots, 2 * N, 20, 2)
    heads, ticks = [0xFFFFFFFF] * n_src, [C.c_uint32(0) for _ in range(n_src)]
    pad = np.zeros(per + 8, np.int16)
    play, far = np.zeros((T, N), np.int16), np.zeros((T, N), np.int16)
    ns_pa = None
    if stages & 16:  # webrtcEnable[WR_NS_PA]: ONE suppressor handle over the played packages, call by call (the FIFO sees its output)
        ns_pa = _fn(lib, "orc_ns_init", C.c_void_p, [C.c_int, C.c_int])(1, 8000)
        ns_run = _fn(lib, "orc_ns_run", None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int])
        assert ns_pa
    zero = np.zeros(N, np.int16)
    near, out = np.zeros((T, n_rec, N), np.int16), np.zeros((T, n_rec, N), np.int16)
    c_open = _fn(lib, "orc_chain_open", C.c_void_p, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint])
    c_step = _fn(lib, "orc_chain_step", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int])
    chains = [c_open(1, 8000, 20, agc_value, stages & 15) for _ in range(n_rec)]
    rw_head, rw_tick, rw_pad = 0xFFFFFFFF, C.c_uint32(0), np.zeros(N + 8, np.int16)
    for t in range(T):
        for i in range(n_src):
            pad[:per] = sources[t, i]
            heads[i] = lib.orc_load_data(C.byref(r), pad.ctypes.data_as(C.c_void_p), per * 2, src_freq, src_chn, 16, C.c_uint32(heads[i]), 1,
                                         C.byref(ticks[i]))
        pos = (r.head_off // 2 + np.arange(N)) % (size // 2)  # the play thread's package: copy out, zero, head and tick move on
        play[t] = ring[pos]
        ring[pos] = 0
        r.head_off = (r.head_off + 2 * N) % size
        r.tick += 2 * N
        if ns_pa:
            ns_run(ns_pa, play[t].ctypes.data, play[t].ctypes.data, N)
        lib.orc_pkgfifo_add(C.byref(f), play[t].ctypes.data_as(C.c_void_p))
        assert lib.orc_pkgfifo_get(C.byref(f), far[t].ctypes.data_as(C.c_void_p), aec_delay_ms) == 0
        near[t] = tick_room(local[t], far[t], far[t - 1] if t else zero)
        out[t] = near[t]
        for k in range(n_rec):  # the record heartbeat of handle set k, in place
            assert c_step(chains[k], far[t].ctypes.data, out[t, k].ctypes.data, N) == 0
        if stages & 32:
            rw_pad[:N] = out[t, 0]
            rw_head = lib.orc_load_data(C.byref(r), rw_pad.ctypes.data_as(C.c_void_p), 2 * N, 8000, 1, 16, C.c_uint32(rw_head), 1, C.byref(rw_tick))
    if ns_pa:
        _fn(lib, "orc_ns_release", None, [C.c_void_p])(ns_pa)
    for c in chains:
        _fn(lib, "orc_chain_close", None, [C.c_void_p])(c)
    zoom = np.stack([np.stack([mix_zoom(lib, 1, 8000, out[t, k], 1, 8000) for k in range(n_rec)]) for t in range(T)])
    return {"play": play, "far": far, "near": near, "out": out, "zoom": zoom}


def tick_ref(sources, local, src_freq, src_chn, stages=15, agc_value=5, platform="alsa"):
    """The same tick composed from the REAL functions (oracle/_ref/ref_mix_driver tick: wmix_load_data, playPkgBuff_add / _get,
    ns_process / aec_process2 / agc_process / vad_process, wmix_pcm_zoom as compiled from /root/reference).  Same arguments and
    result as tick_port (no `near`).  platform: which of the reference's platform builds (AEC_INTERVALMS and VIEW_PLAY_CORRECT are
    compile-time there)."""
    T, n_src, per = sources.shape
    n_rec, N = local.shape[1], 160
    blob = b"".join(np.ascontiguousarray(sources[t]).tobytes() + np.ascontiguousarray(local[t]).tobytes() for t in range(T))
    raw = np.frombuffer(ref_mix("tick", n_src, src_freq, src_chn, n_rec, T, stages, agc_value, stdin=blob, platform=platform), np.int16)
    raw = raw.reshape(T, 2 + 2 * n_rec, N)
    return {"play": raw[:, 0], "far": raw[:, 1], "out": raw[:, 2::2], "zoom": raw[:, 3::2]}


# ---------------------------------------------------------------- reference mixer (executable)
# the reference's three platform builds (platform/<name>/plat.h:10-21): (PLAT_AEC_INTERVALMS, PLAT_PLAY_CORRECT in bytes).  The ring
# is 1 x 8000 Hz in all of them; oracle/Makefile builds src/wmix.c and the driver once per header.
PLATFORMS = {"alsa": (400, 3200), "hi3516": (700, 0), "t31": (0, 0)}


def ref_mix_exe(platform="alsa"):
    assert platform in PLATFORMS, platform
    return REF_MIX if platform == "alsa" else REF_MIX + "_" + platform


def ref_mix(*args, stdin=b"", platform="alsa"):
    return subprocess.run([ref_mix_exe(platform)] + [str(a) for a in args], input=stdin, stdout=subprocess.PIPE, check=True).stdout


# ---------------------------------------------------------------- hashing for golden files
def fnv1a64(b):
    h = 0xCBF29CE484222325
    for x in bytes(b):
        h = ((h ^ x) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h
