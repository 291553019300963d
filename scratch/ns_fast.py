import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
import numpy as np, torch
from oracle import loader as L
from make_ns_golden import ns_case_input
from test_ns_gpu import run_gpu
port=L.port(); cuda=torch.device('cuda:0')
for chn,freq in ((1,16000),(1,8000)):
    S,nf=64,1100
    x=np.stack([ns_case_input(chn,freq,nf,seed=1000+31*s) for s in range(S)])
    want=np.stack([L.run_ns(port,chn,freq,x[s],freq//100,prefix='orc') for s in range(S)])
    fast=run_gpu(cuda,chn,freq,x,ordered=False,packets_per_launch=100)
    d=np.abs(fast.astype(int)-want.astype(int))
    per=freq//100*chn
    print(chn,freq,'max',d.max(),'n>0',int((d>0).sum()),'n>1',int((d>1).sum()),'of',d.size)
    bad=np.argwhere(d>1)
    if len(bad):
        ss=sorted(set(bad[:,0])); print('streams with >1:',ss)
        for s in ss[:4]:
            fr=np.nonzero((d[s].reshape(nf,per)>1).any(1))[0]
            print(' stream',s,'frames',fr[:10],'...',fr[-3:],'count',len(fr),'max',d[s].max(), 'rms',np.sqrt((d[s]**2).mean()))
