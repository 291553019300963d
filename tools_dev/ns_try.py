import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import loader as L
from wmix_amd import synth
from wmix_amd.ns import NsBatch
port=L.port()
dev=torch.device('cuda:0')
for chn,freq in ((1,16000),(1,8000),(2,16000),(1,32000),(2,32000),(2,8000)):
    pkt=freq//100; S=8; NF=int(sys.argv[1]) if len(sys.argv)>1 else 600
    x=np.stack([synth.ns_input(100+s*17,chn,NF,pkt).T.reshape(-1) for s in range(S)]).astype(np.int16)  # [S, NF*pkt*chn]
    x[3].reshape(NF,-1)[60:70]=0
    want=np.stack([L.run_ns(port,chn,freq,x[s],pkt,prefix='orc') for s in range(S)])
    for ordered in (True,):
        nb=NsBatch(S,chn,freq)
        d=torch.from_numpy(x.reshape(S,NF,pkt*chn).copy()).to(dev)
        t0=time.time()
        for f in range(0,NF,50):   # several launches of 50 packets
            nb.process(d[:,f:f+50])
        torch.cuda.synchronize(); t1=time.time()
        got=d.cpu().numpy().reshape(S,-1)
        diff=np.abs(got.astype(int)-want.astype(int))
        first=[int(np.argmax(diff[s]>0))//(pkt*chn) if diff[s].any() else -1 for s in range(S)]
        print(chn,freq,'ordered' if ordered else 'fast','maxdiff',diff.max(),'ndiff',int((diff>0).sum()),'of',diff.size,'first bad frame per stream',first,'%.2fs'%(t1-t0), flush=True)
        nb.close()
