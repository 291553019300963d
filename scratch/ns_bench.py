import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from wmix_amd import synth
from wmix_amd.ns import NsBatch
dev=torch.device('cuda:0')
for S,freq,ordered in ((4096,16000,True),(4096,16000,False),(65536,16000,True),(65536,16000,False),(65536,8000,True)):
    pkt=freq//100
    x=synth.ns_input(1,256,40,pkt)  # [256, 40*pkt]
    d=torch.from_numpy(np.tile(x.reshape(256,40,pkt),(S//256,1,1)).transpose(1,0,2).copy()).to(dev)  # [40,S,pkt] packet-major
    nb=NsBatch(S,1,freq,ordered=ordered)
    for f in range(10): nb.process_packet_major(d[f:f+1])
    torch.cuda.synchronize(); t0=time.time()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for f in range(10,40): nb.process_packet_major(d[f:f+1])
    e1.record(); torch.cuda.synchronize(); t1=time.time()
    ms=e0.elapsed_time(e1)/30
    print(S,freq,'ordered' if ordered else 'fast','%.3f ms/step'%ms,'%.3e frames/s'%(S/ms*1e3), 'hbm frac %.4f'%(S*25040/ms*1e3/8e12), flush=True)
    nb.close()
