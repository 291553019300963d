import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
import numpy as np, torch, time
from oracle import loader as L
from wmix_amd import synth
from test_ns_gpu import run_gpu
from test_aec_gpu import gpu_aec, gpu_chain, check_float_path
from make_ns_golden import ns_case_input
port=L.port(); cuda=torch.device('cuda:0')
for chn,freq in ((1,16000),(2,32000),(1,8000),(2,16000)):
    S,nf=192,1600
    x=np.stack([ns_case_input(chn,freq,nf,seed=50000+17*s) for s in range(S)])
    got=run_gpu(cuda,chn,freq,x,packets_per_launch=100,packet_major=(chn==1))
    bad=0
    for s in range(0,S,3):
        want=L.run_ns(port,chn,freq,x[s],freq//100,prefix='orc')
        bad+=int((got[s]!=want).sum())
    print('NS',chn,freq,'mismatching samples',bad,flush=True)
for freq in (16000,8000):
    S,n=160,2000; pkg=freq//100
    far=synth.far_end(91000+freq,n,pkg); near=synth.near_end(92000+freq,S,n,pkg,far=far)
    got=gpu_aec(cuda,1,freq,10,0,far,near,pkts_per_launch=50,packet_major=True)
    nd=0; mx=0
    for s in range(0,S,4):
        want=L.run_aec(port,1,freq,10,far,near[s],pkg,0,prefix='orc')
        d=np.abs(got[s].astype(int)-want.astype(int)); nd+=int((d>0).sum()); mx=max(mx,int(d.max()))
    print('AEC',freq,'differing samples',nd,'max',mx,flush=True)
S,n=96,2500
far=synth.far_end(93000,n,160); near=synth.near_end(93100,S,n,160,far=far)
got=gpu_chain(cuda,1,16000,15,far,near,pkts_per_launch=50)
nd=0; mx=0
for s in range(0,S,3):
    want=L.run_chain(port,1,16000,5,15,far,near[s],160,prefix='orc')
    d=np.abs(got[s].astype(int)-want.astype(int)); nd+=int((d>0).sum()); mx=max(mx,int(d.max()))
print('CHAIN differing',nd,'max',mx)
# ---- fixed-point alternates (round 2): longer and more varied than tests/test_nsx_gpu.py / test_aecm_gpu.py
import test_nsx_gpu as TN, test_aecm_gpu as TA
from make_nsx_golden import nsx_case_input
from make_aecm_golden import aecm_case_input
rng = np.random.default_rng(12345)
for chn, freq in ((1, 16000), (2, 32000), (1, 8000), (2, 16000), (2, 8000), (1, 32000)):
    S, nf = 192, 2200  # four 512-block threshold updates
    amps = rng.choice([1, 7, 60, 500, 2500, 9000, 20000, 32000], size=S)
    x = np.stack([nsx_case_input(chn, freq, nf, int(amps[s]), seed=70000 + 13 * s) for s in range(S)])
    for s in range(0, S, 11):  # silences and full-scale bursts of random length
        a, b = sorted(rng.integers(0, nf, 2)); x[s].reshape(nf, -1)[a:b] = 0
    for s in range(5, S, 17):
        a = int(rng.integers(0, nf - 50)); x[s].reshape(nf, -1)[a:a + 40] = rng.choice([-32768, 32767], size=x[s].reshape(nf, -1)[a:a + 40].shape)
    got = TN.run_gpu(cuda, chn, freq, x, packets_per_launch=int(rng.integers(1, 300)), packet_major=(chn == 1))
    bad = 0
    for s in range(0, S, 3):
        bad += int((got[s] != L.run_nsx(port, chn, freq, x[s], freq // 100, prefix='orc')).sum())
    print('NSX', chn, freq, 'mismatching samples', bad, flush=True)
for chn, freq, iv in ((1, 16000, 10), (1, 8000, 10), (2, 16000, 10), (1, 8000, 20), (2, 8000, 10)):
    S, n = 160, 2500
    far, _, pkt = aecm_case_input(chn, freq, iv, n, seed=81000 + freq + iv)
    f0 = far[::chn].astype(np.int32)
    near = np.zeros((S, n * pkt * chn), np.int16)
    for s in range(S):
        delay, gain = int(rng.integers(1, 2500)), float(rng.uniform(0.02, 1.5))
        echo = np.zeros_like(f0); echo[delay:] = (f0[:-delay] * gain).astype(np.int32)
        xx = echo + synth.lcg_noise([9000 + s], n * pkt, int(rng.integers(1, 2000)))[0].astype(np.int32)
        if s % 4 == 0:
            xx += np.trunc(synth.gated_tone(n, pkt, amp=float(rng.uniform(100, 20000)), period=int(rng.integers(5, 400)))).astype(np.int32)
        if s % 13 == 0:
            a, b = sorted(rng.integers(0, xx.size, 2)); xx[a:b] = 0
        xx = np.clip(xx, -32768, 32767).astype(np.int16)
        near[s] = np.repeat(xx, chn) if chn == 2 else xx
    got, rc = TA.run_gpu(cuda, chn, freq, iv, far, near, packets_per_launch=int(rng.integers(1, 32)), packet_major=(chn == 1))
    bad = 0
    for s in range(0, S, 4):
        bad += int((got[s] != L.run_aecm(port, chn, freq, iv, far, near[s], pkt, prefix='orc')).sum())
    print('AECM', chn, freq, iv, 'rc', rc, 'mismatching samples', bad, flush=True)
