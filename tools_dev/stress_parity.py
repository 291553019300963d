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
