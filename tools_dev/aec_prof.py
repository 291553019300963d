# developer tool: per-phase cycle split of aec_near_kernel (library built with make EXTRA=-DWMX_AEC_PROF)
import sys, ctypes; sys.path.insert(0,'.')
import numpy as np, torch
from wmix_amd import synth, _lib
from wmix_amd.aec import AecBatch
dev=torch.device('cuda:0'); S=int(sys.argv[1]) if len(sys.argv)>1 else 65536
import os
if os.environ.get('WMX_TOOL_LIB'): _lib.LIB_PATH=os.environ['WMX_TOOL_LIB']  # profiling build kept beside the product library
lib=_lib.lib(); f=lib.wmx_debug_aec_prof; f.argtypes=[ctypes.c_void_p,ctypes.c_int]
ab=AecBatch(S,1,16000,10); nf=24
far=synth.far_end(5,nf,160); near=synth.near_end(50,64,nf,160,far=far)
dfar=torch.from_numpy(far.reshape(nf,160).copy()).to(dev)
dn=torch.from_numpy(np.ascontiguousarray(near.reshape(64,nf,160).transpose(1,0,2))).to(dev).repeat(1,S//64,1).contiguous()  # [packet][stream][160], as bench.py lays a batch out
buf=(ctypes.c_ulonglong*16)()
for k in range(nf):
    if k==16: f(buf,1)
    ab.process2_packet_major(dfar[k:k+1],dn[k:k+1])
f(buf,0); v=np.array(buf[:16],dtype=np.float64)
names=['near ring+d,dw fft','dpow+filterfar','y ifft+e,ew fft','scale+adapt','partdelay','xfw+psd','sd/se sums','coh+hNl+scalars','overdrive+cn','ifft+ola','state in','(all pkts)','state out','scale_err','pack','24 ffts']
tot=v[10]+v[12]+v[11]
for n,x in zip(names,v): print('%-20s %8.1f Mcyc %5.1f%%'%(n,x/1e6,100*x/tot))
print('pcm io etc', (v[11]-v[:10].sum()-v[13:].sum())/1e6)
