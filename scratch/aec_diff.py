import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
import numpy as np, torch
from oracle import loader as L
from wmix_amd import synth
from test_aec_gpu import gpu_aec
port=L.port(); cuda=torch.device('cuda:0')
for freq in (16000,8000):
    S,n=96,1300; pkg=freq//100
    far=synth.far_end(6001,n,pkg); near=synth.near_end(6100,S,n,pkg,far=far); near[7]=0; near[8,40:]=far[:-40]//2
    want=np.stack([L.run_aec(port,1,freq,10,far,near[s],pkg,0,prefix='orc') for s in range(S)])
    got=gpu_aec(cuda,1,freq,10,0,far,near,pkts_per_launch=50,packet_major=True)
    d=np.abs(got.astype(int)-want.astype(int))
    bad=np.argwhere(d>0)
    print(freq,'max',d.max(),'n',len(bad),'of',d.size)
    for s in sorted(set(bad[:,0]))[:6]:
        idx=bad[bad[:,0]==s][:,1]
        print('  stream',s,'count',len(idx),'pkts',sorted(set(idx//pkg))[:12])
