# developer tool: per-phase cycle split of ns_kernel (library built with make EXTRA=-DWMX_NS_PROF)
import sys, ctypes; sys.path.insert(0,'.')
import numpy as np, torch
from wmix_amd import synth, _lib
from wmix_amd.ns import NsBatch
dev=torch.device('cuda:0'); S=int(sys.argv[1]) if len(sys.argv)>1 else 65536
import os
if os.environ.get('WMX_TOOL_LIB'): _lib.LIB_PATH=os.environ['WMX_TOOL_LIB']  # profiling build kept beside the product library
lib=_lib.lib(); f=lib.wmx_debug_ns_prof; f.argtypes=[ctypes.c_void_p,ctypes.c_int]
nb=NsBatch(S,1,16000); nf=int(sys.argv[2]) if len(sys.argv)>2 else 260
x=synth.ns_input(7,64,nf,160)
d=torch.from_numpy(np.ascontiguousarray(x.reshape(64,nf,160).transpose(1,0,2))).to(dev).repeat(1,S//64,1).contiguous()  # [packet][stream][160]
buf=(ctypes.c_ulonglong*16)()
for k in range(nf):
    if k==nf-8: f(buf,1)
    nb.process_packet_major(d[k:k+1])
f(buf,0); v=np.array(buf[:14],dtype=np.float64)
names=['load+window+energy1','fft fwd','spectrum+log','7 sums','noise est (quantiles)','startup/snr loop','3 sums+flat','lrt loop (log)','ksum+prior+exp loop','noise update+wiener','ifft','td+energy2','synth+hb','output']
for n,x in zip(names,v): print('%-24s %8.1f Mcyc %5.1f%%'%(n,x/1e6,100*x/v.sum()))
