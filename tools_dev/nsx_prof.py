# developer tool: per-phase cycle split of nsx_kernel (library built with make EXTRA=-DWMX_NSX_PROF, WMX_TOOL_LIB=that build)
import ctypes
import os
import sys

sys.path.insert(0, '.')
import numpy as np
import torch

from wmix_amd import _lib, synth

if os.environ.get('WMX_TOOL_LIB'):
    _lib.LIB_PATH = os.environ['WMX_TOOL_LIB']
from wmix_amd.nsx import NsxBatch

dev = torch.device('cuda:0')
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 260
lib = _lib.lib()
f = lib.wmx_debug_nsx_prof
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
nb = NsxBatch(S, 1, 16000)
x = synth.ns_input(7, 64, nf, 160)
d = torch.from_numpy(np.ascontiguousarray(x.reshape(64, nf, 160).transpose(1, 0, 2))).to(dev).repeat(1, S // 64, 1).contiguous()  # [packet][stream][160]: the batch's packets lie side by side (a stream-major array puts every stream's packet on its own page)
work = torch.empty_like(d[0:1])
buf = (ctypes.c_ulonglong * 16)()
for k in range(nf):
    if k == nf - 8:
        f(buf, 1)
    nb.process_packet_major(d[k:k + 1], work)
f(buf, 0)
v = np.array(buf[:15], dtype=np.float64)
names = ['spectrum, magnitudes, sums', 'spectral flatness', 'noise estimation (quantiles)', 'start-up blend', 'step 1 prior / post snr',
         'spectral difference', 'feature extraction', 'speech / noise probability', 'noise update', 'step 3 wiener gain', 'synthesis behind the inverse fft', 'packet shifted in (global loads)', 'window, energy, max', 'forward fft',
         'spectrum filtered + inverse fft']
for n, c in zip(names, v):
    print('%-36s %8.1f Mcyc %5.1f%%' % (n, c / 1e6, 100 * c / v.sum()))
