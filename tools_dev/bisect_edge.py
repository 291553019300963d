import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from oracle import loader as L
from test_aec_gpu import gpu_chain
port=L.port()
cuda=torch.device('cuda:0')
freq=int(sys.argv[1]) if len(sys.argv)>1 else 16000
n=3000; pkt=freq//100
rng=np.random.default_rng(77)
t=np.arange(n*pkt)
noise=lambda a: rng.integers(-a,a+1,n*pkt).astype(np.int16)
far=noise(8000); far[60*pkt:]=0
near=np.where((t//40)%2==0,32767,-32768).astype(np.int16)[None,:]
for stages in (1,2,4,8,3,7,15):
    got=gpu_chain(cuda,1,freq,stages,far,near)
    want=L.run_chain(port,1,freq,5,stages,far,near[0],pkt,prefix="orc")
    d=np.abs(got[0].astype(int)-want.astype(int))
    nz=np.nonzero(d)[0]
    print("stages",stages,"max diff",d.max(),"n diff",len(nz),"first idx",(nz[:3], nz[:1]//pkt) if len(nz) else None, flush=True)
