# developer tool: per-phase cycle split of aecm_near_kernel (library built with make EXTRA=-DWMX_AECM_PROF, WMX_TOOL_LIB=that build)
import ctypes
import os
import sys

sys.path.insert(0, '.')
import numpy as np
import torch

from wmix_amd import _lib, synth

if os.environ.get('WMX_TOOL_LIB'):
    _lib.LIB_PATH = os.environ['WMX_TOOL_LIB']
from wmix_amd.aecm import AecmBatch

dev = torch.device('cuda:0')
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 260
lib = _lib.lib()
f = lib.wmx_debug_aecm_prof
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
ab = AecmBatch(S, 1, 16000, 10)
far = synth.far_end(3000, nf, 160)
near = synth.near_end(3001, 64, nf, 160, far=far).reshape(64, nf, 160)
dfar = torch.from_numpy(far.reshape(nf, 160).copy()).to(dev)
d = torch.from_numpy(np.ascontiguousarray(near.transpose(1, 0, 2))).to(dev).repeat(1, S // 64, 1).contiguous()  # [packet][stream][160]
work = torch.empty_like(d[0:1])
buf = (ctypes.c_ulonglong * 16)()
for k in range(nf):
    if k == nf - 8:
        f(buf, 1)
    work.copy_(d[k:k + 1])
    ab.process2_packet_major(dfar[k:k + 1], work)
f(buf, 0)
v = np.array(buf[:10], dtype=np.float64)
names = ['near spectrum (window, fft, magnitudes)', 'delay estimator (binary spectra)', 'aligned far spectrum', 'energies', 'step size', 'channel update',
         'suppression gain', 'wiener filter', 'comfort noise', 'inverse fft + window']
for n, c in zip(names, v):
    print('%-40s %8.1f Mcyc %5.1f%%' % (n, c / 1e6, 100 * c / v.sum()))
