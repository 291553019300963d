import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import loader as L
from wmix_amd import synth
from wmix_amd.aec import AecBatch
port=L.port(); dev=torch.device('cuda:0')
NF=int(sys.argv[1]) if len(sys.argv)>1 else 600
for chn,freq,ims,delay in ((1,16000,10,0),(1,8000,10,0),(1,8000,20,0),(2,16000,10,0),(1,16000,10,120)):
    pkg=freq//1000*(20 if (freq<=8000 and ims%20==0) else 10); S=6
    nf=NF*(freq//100)//pkg
    far=synth.far_end(5,nf,pkg)
    near=synth.near_end(50,S,nf,pkg,far=far)
    near[2]=0
    if chn==2:
        far2=np.stack([far,far//2],1).reshape(-1); near2=np.stack([near,near//3],2).reshape(S,-1)
    else: far2,near2=far,near
    want=np.stack([L.run_aec(port,chn,freq,ims,far2,near2[s],pkg,delay,prefix='orc') for s in range(S)])
    ab=AecBatch(S,chn,freq,ims)
    dfar=torch.from_numpy(far2.reshape(nf,pkg*chn).copy()).to(dev)
    dn=torch.from_numpy(near2.reshape(S,nf,pkg*chn).copy()).to(dev)
    t0=time.time()
    for f in range(0,nf,37):
        rc,_=ab.process2(dfar[f:f+37],dn[:,f:f+37],delay_ms=delay)
        assert rc==0
    torch.cuda.synchronize(); t1=time.time()
    got=dn.cpu().numpy().reshape(S,-1)
    d=np.abs(got.astype(int)-want.astype(int))
    per=pkg*chn
    first=[int(np.argmax(d[s]>0))//per if d[s].any() else -1 for s in range(S)]
    print(chn,freq,ims,delay,'maxdiff',d.max(),'ndiff',int((d>0).sum()),'of',d.size,'first bad pkt',first,'%.2fs'%(t1-t0),flush=True)
    ab.close()
