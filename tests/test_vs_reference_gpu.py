"""The kernels against the REAL reference on the GPU box -- oracle/_ref/libwmixref.so, the reference's own sources compiled by
oracle/Makefile where /root/reference exists; the built library travels with the repository -- with no restatement in between:
the synthetic recipe on many streams, the extreme signals of tests/test_extremes_gpu.py, every module on its own and the whole
chain.  (The other GPU tests compare with oracle/orc_*.c, which the CPU suite pins on this same library.)"""
import numpy as np
import pytest

from oracle import loader as L
from wmix_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("freq", [16000, 8000])
def test_chain_many_streams_vs_the_real_reference(cuda, oracle_ref, freq):
    from test_aec_gpu import check_float_path, gpu_chain
    S, n, pkt = 24, 500, freq // 100
    far = synth.far_end(8100 + freq // 8000, n, pkt)
    near = synth.near_end(8200 + freq // 8000, S, n, pkt, far=far).reshape(S, n * pkt)
    got = gpu_chain(cuda, 1, freq, 15, far, near.copy())
    want = np.stack([L.run_chain(oracle_ref, 1, freq, 5, 15, far, near[s], pkt, prefix="ref") for s in range(S)])
    check_float_path(got, want, max_fraction=1e-4)


@pytest.mark.parametrize("freq", [8000, 16000, 32000])
def test_every_module_on_extreme_signals_vs_the_real_reference(cuda, oracle_ref, freq):
    from test_aec_gpu import check_float_path, gpu_aec
    from test_aecm_gpu import run_gpu as gpu_aecm
    from test_extremes_gpu import extreme_signals
    from test_ns_gpu import run_gpu as gpu_ns
    from test_nsx_gpu import run_gpu as gpu_nsx
    from test_vadagc_gpu import gpu_agc, gpu_vad
    pkt, names, x = extreme_signals(freq, 400)
    assert np.array_equal(gpu_agc(cuda, 1, freq, 5, x.copy()), np.stack([L.run_agc(oracle_ref, 1, freq, 5, s, pkt, prefix="ref") for s in x]))
    assert np.array_equal(gpu_vad(cuda, 1, freq, 10, 1, x.copy()), np.stack([L.run_vad(oracle_ref, 1, freq, 10, s, pkt, prefix="ref") for s in x]))
    assert np.array_equal(gpu_ns(cuda, 1, freq, x.copy()), np.stack([L.run_ns(oracle_ref, 1, freq, s, pkt, prefix="ref") for s in x]))
    if freq == 32000:
        return
    far = x[names.index("loud")]
    assert np.array_equal(gpu_nsx(cuda, 1, freq, x.copy()), np.stack([L.run_nsx(oracle_ref, 1, freq, s, pkt, prefix="ref") for s in x]))
    got, rc = gpu_aecm(cuda, 1, freq, 10, far, x.copy())
    assert rc == 0 and np.array_equal(got, np.stack([L.run_aecm(oracle_ref, 1, freq, 10, far, s, pkt, prefix="ref") for s in x]))
    check_float_path(gpu_aec(cuda, 1, freq, 10, 0, far, x.copy()), np.stack([L.run_aec(oracle_ref, 1, freq, 10, far, s, pkt, prefix="ref") for s in x]),
                     max_fraction=1e-4)
