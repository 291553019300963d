"""Parity at the BASELINE.json configurations that are not bench lines:
configs[3]  NS + AEC, 8 kHz mono, shared far-end (131 072 streams per GPU when the 1 M streams shard over 8 GPUs)
configs[4]  2-channel 32 kHz NS + AGC, then wmix_load_data's resample to the 8 kHz mono ring with an N-way mix.
Small cases against the oracle sample for sample; the full sizes through properties plus an oracle spot check."""
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
from oracle import loader  # noqa: E402
from test_aec_gpu import check_float_path, gpu_chain  # noqa: E402
from test_mix_oracle import _bind, orc_load  # noqa: E402
from wmix_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu


def test_config3_ns_aec_8k_vs_oracle(cuda, oracle_port):
    S, n, pkt = 48, 400, 80
    far = synth.far_end(31, n, pkt)
    near = synth.near_end(32, S, n, pkt, far=far)
    near[5] = 0  # a silent stream takes the zero-energy early-outs
    got = gpu_chain(cuda, 1, 8000, 3, far, near)
    for s in range(0, S, 5):
        want = loader.run_chain(oracle_port, 1, 8000, 5, 3, far, near[s], pkt, prefix="orc")
        check_float_path(got[s], want)


def test_config3_full_size(cuda, oracle_port):
    """131 072 streams of 8 kHz NS -> AEC in one batch: streams with equal input give equal output wherever they
    sit, and sampled streams agree with the oracle."""
    from wmix_amd.aec import AecBatch
    from wmix_amd.ns import NsBatch
    S, n, pkt, U = 131072, 24, 80, 64
    far = synth.far_end(41, n, pkt)
    uniq = synth.near_end(42, U, n, pkt, far=far).reshape(U, n, pkt)
    d = torch.from_numpy(uniq).to(cuda).repeat(S // U, 1, 1).contiguous()  # stream s carries input s % U
    dfar = torch.from_numpy(far.reshape(n, pkt).copy()).to(cuda)
    ns, aec = NsBatch(S, 1, 8000), AecBatch(S, 1, 8000, 10)
    for f in range(0, n, 8):
        ns.process(d[:, f:f + 8])
        rc, _ = aec.process2(dfar[f:f + 8], d[:, f:f + 8])
        assert rc == 0
    first = d[:U]
    assert torch.equal(d.view(S // U, U, n, pkt), first.expand(S // U, U, n, pkt))
    got = first.cpu().numpy().reshape(U, -1)
    for s in (0, 17, 63):
        want = loader.run_chain(oracle_port, 1, 8000, 5, 3, far, uniq[s].reshape(-1), pkt, prefix="orc")
        check_float_path(got[s], want)
    ns.close()
    aec.close()


def test_config2_full_size_chain(cuda, oracle_port):
    """configs[2] at full size: 65 536 streams of 16 kHz NS -> AEC -> AGC -> VAD in one batch, one packet per launch like
    bench.py.  Streams with equal input give equal output wherever they sit, and sampled streams agree with the oracle
    chain (round-1 VERDICT: the full size used to be exercised stage by stage only, NS never at 65 536)."""
    from wmix_amd.aec import AecBatch
    from wmix_amd.agc import AgcBatch
    from wmix_amd.ns import NsBatch
    from wmix_amd.vad import VadBatch
    S, n, pkt, U = 65536, 30, 160, 64
    far = synth.far_end(51, n, pkt)
    uniq = synth.near_end(52, U, n, pkt, far=far).reshape(U, n, pkt)
    uniq[7] = 0
    d = torch.from_numpy(uniq).to(cuda)[torch.arange(S, device=cuda) % U].transpose(0, 1).contiguous()  # [n, S, pkt]
    dfar = torch.from_numpy(far.reshape(n, pkt).copy()).to(cuda)
    ns, aec, agc, vad = NsBatch(S, 1, 16000), AecBatch(S, 1, 16000, 10), AgcBatch(S, 1, 16000, 5), VadBatch(S, 1, 16000, 10)
    for f in range(n):
        ns.process_packet_major(d[f:f + 1])
        rc, _ = aec.process2_packet_major(dfar[f:f + 1], d[f:f + 1])
        assert rc == 0
        agc.process_packet_major(d[f:f + 1])
        vad.process_packet_major(d[f:f + 1])
    first = d[:, :U]
    assert torch.equal(d.view(n, S // U, U, pkt), first.unsqueeze(1).expand(n, S // U, U, pkt))
    got = first.transpose(0, 1).cpu().numpy().reshape(U, -1)
    for s in (0, 7, 29, 63):
        want = loader.run_chain(oracle_port, 1, 16000, 5, 15, far, uniq[s].reshape(-1), pkt, prefix="orc")
        check_float_path(got[s], want)
    for b in (ns, aec, agc, vad):
        b.close()


def _ns_agc_32k_2ch(cuda, src):
    """src int16 [S, n_pkts, 640] (10 ms of 2 x 32 kHz) -> NS then AGC in place, like the daemon's record chain with
    AEC and VAD switched off (src/wmix.c:613-709); the AGC works in 5 ms packets at 32 kHz (src/webrtc.c:724-727)."""
    from wmix_amd.agc import AgcBatch
    from wmix_amd.ns import NsBatch
    S, n, per = src.shape
    d = torch.from_numpy(src).to(cuda)
    ns, agc = NsBatch(S, 2, 32000), AgcBatch(S, 2, 32000, 5)
    assert ns.pkt == 640 and agc.pkt == 320
    ns.process(d)
    agc.process(d.view(S, 2 * n, 320))
    ns.close()
    agc.close()
    return d


def test_config4_ns_agc_resample_mix_vs_oracle(cuda, oracle_port):
    from wmix_amd.mix import MixBatch
    _bind(oracle_port)
    G_, N, n = 3, 4, 30  # mix groups, sources per group, 10 ms packets
    S, per = G_ * N, 640
    rng = np.random.default_rng(77)
    t = np.arange(n * 320)
    src = np.zeros((S, n * 320, 2), np.int16)
    for s in range(S):
        tone = 9000 * np.sin(2 * np.pi * (200 + 37 * s) * t / 32000) * (((t // 3200) + s) % 3 > 0)
        src[s, :, 0] = np.clip(tone + rng.integers(-1500, 1500, t.size), -32768, 32767)
        src[s, :, 1] = src[s, :, 0] // 3  # wmix hands the R channel to NS as the "high band" 
    src = src.reshape(S, n, per)
    d = _ns_agc_32k_2ch(cuda, src.copy())
    # oracle: the same two stages per source
    want_pcm = np.stack([loader.run_chain(oracle_port, 2, 32000, 5, 1 | 4, np.zeros(n * per, np.int16), src[s].reshape(-1), 320, prefix="orc")
                         for s in range(S)]).reshape(S, n, per)
    assert np.array_equal(d.cpu().numpy(), want_pcm)
    # every 10 ms: the N sources of a group are resampled (2 x 32 kHz -> 1 x 8 kHz, L channel, every 4th frame) and
    # accumulated with saturation into the group's ring in call order; then the play thread drains 10 ms
    mb = MixBatch(G_, 1, 8000)
    pad = torch.zeros(S, n, 2, dtype=torch.int16, device=cuda)
    dsrc = torch.cat([d, pad], 2).view(G_, N, n, per + 2)
    for k in range(n):
        mb.set(0, 0, 1)
        h, tk = mb.load(dsrc[:, :, k].contiguous(), per * 2, 32000, 2)
        mb.set(3200, 0, 1)
        out = mb.drain(160).cpu().numpy()
        for g in range(G_):
            flat = np.concatenate([want_pcm[g * N + i, k] for i in range(N)] + [np.zeros(2, np.int16)])
            ring, meta = orc_load(oracle_port, 1, 8000, 32000, 2, 1, 1, N, per * 2, 0, flat)
            assert np.array_equal(out[g], ring[1600:1680]), (k, g)
            assert (h, tk) == (int(meta[-1][1]), int(meta[-1][0]))
    mb.close()


def test_config4_full_size_32768_sources(cuda, oracle_port):
    """BASELINE configs[4] at its real size: 32 768 two-channel 32 kHz sources through
    ns_kernel<256, 2> and the two-channel AGC pipeline, then the 8-way resample-and-mix into 4 096 rings of 1 x 8 kHz
    with the play thread's 10 ms drain -- packet-major like bench.py.  64 distinct sources replicated over the batch: equal
    input must give equal output wherever a source sits; spot sources and spot mix groups agree with the oracle."""
    from wmix_amd.agc import AgcBatch
    from wmix_amd.mix import MixBatch
    from wmix_amd.ns import NsBatch
    _bind(oracle_port)
    S, N, n, per, U = 32768, 8, 12, 640, 64
    rng = np.random.default_rng(404)
    t = np.arange(n * 320)
    uniq = np.zeros((U, n * 320, 2), np.int16)
    for s in range(U):
        tone = 14000 * np.sin(2 * np.pi * (150 + 31 * s) * t / 32000) * (((t // 1600) + s) % 4 > 0)
        uniq[s, :, 0] = np.clip(tone + rng.integers(-2500, 2500, t.size), -32768, 32767)
        uniq[s, :, 1] = uniq[s, :, 0] // 3
    uniq = uniq.reshape(U, n, per)
    uniq[5] = 0  # a silent source inside every group
    inp = torch.from_numpy(uniq).to(cuda)[torch.arange(S, device=cuda) % U].transpose(0, 1).contiguous()  # [n, S, per]
    flat = torch.zeros(S * per + 2, dtype=torch.int16, device=cuda)  # + the mixer's look-ahead frame
    work = flat[: S * per].view(1, S, per)
    src = torch.as_strided(flat, (S // N, N, per + 2), (N * per, per, 1))
    ns, agc, mb = NsBatch(S, 2, 32000), AgcBatch(S, 2, 32000, 5), MixBatch(S // N, 1, 8000)
    pcm, mixes = [], []
    for k in range(n):
        ns.process_packet_major(inp[k:k + 1], work)
        agc.process(work[0].view(S, 2, 320))
        pcm.append(work[0, :U].cpu().numpy())
        assert torch.equal(work.view(S // U, U, per), work[0, :U].unsqueeze(0).expand(S // U, U, per)), k
        mb.set(0, 0, 1)
        mb.load(src, per * 2, 32000, 2)
        mb.set(3200, 0, 1)
        out = mb.drain(160)
        mixes.append(out[: U // N].cpu().numpy())
        # groups repeat with period U / N = 8 (group g holds sources 8g .. 8g + 7 of the replicated set)
        assert torch.equal(out.view(S // U, U // N, -1), out[: U // N].unsqueeze(0).expand(S // U, U // N, out.shape[-1])), k
    pcm = np.stack(pcm, 1)  # [U, n, per]
    for s in (0, 5, 17, 63):
        want = loader.run_chain(oracle_port, 2, 32000, 5, 1 | 4, np.zeros(n * per, np.int16), uniq[s].reshape(-1), 320, prefix="orc")
        assert np.array_equal(pcm[s].reshape(-1), want), s
    for k in (0, 5, n - 1):
        for g in (0, 3, 7):
            fl = np.concatenate([pcm[g * N + i, k] for i in range(N)] + [np.zeros(2, np.int16)])
            ring, _ = orc_load(oracle_port, 1, 8000, 32000, 2, 1, 1, N, per * 2, 0, fl)
            assert np.array_equal(mixes[k][g], ring[1600:1680]), (k, g)
    for b in (ns, agc, mb):
        b.close()
