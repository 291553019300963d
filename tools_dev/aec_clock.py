import sys, ctypes, os; sys.path.insert(0,'.')
import numpy as np, torch
from wmix_amd import synth, _lib
_lib.LIB_PATH=os.environ['WMX_TOOL_LIB']
from wmix_amd.aec import AecBatch
dev=torch.device('cuda:0'); S=65536
lib=_lib.lib(); f=lib.wmx_debug_aec_prof; f.argtypes=[ctypes.c_void_p,ctypes.c_int]
ab=AecBatch(S,1,16000,10); nf=200
far=synth.far_end(5,nf,160); near=synth.near_end(50,64,nf,160,far=far)
dfar=torch.from_numpy(far.reshape(nf,160).copy()).to(dev)
dn=torch.from_numpy(near.reshape(64,nf,160).copy()).to(dev).repeat(S//64,1,1).contiguous()
buf=(ctypes.c_ulonglong*16)()
for k in range(nf):
    if k==nf-40: f(buf,1)
    ab.process2(dfar[k:k+1],dn[:,k:k+1])
f(buf,0); v=np.array(buf[:16],dtype=np.float64)
print("cycles", v[14], "realtime ticks", v[15], "GHz", v[14]/(v[15]*10))
