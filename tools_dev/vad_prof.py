# developer tool: per-phase cycle split of vad_kernel (library built with make EXTRA=-DWMX_VAD_PROF, WMX_TOOL_LIB=that build)
import sys, ctypes, os; sys.path.insert(0, '.')
import numpy as np, torch
from wmix_amd import synth, _lib
if os.environ.get('WMX_TOOL_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['WMX_TOOL_LIB'])
from wmix_amd.vad import VadBatch
dev = torch.device('cuda:0'); S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
freq = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
lib = _lib.lib(); f = lib.wmx_debug_vad_prof; f.argtypes = [ctypes.c_void_p, ctypes.c_int]
pkt = freq // 100
vb = VadBatch(S, 1, freq, 10); nf = 120
x = synth.ns_input(7, 64, nf, pkt)
d = torch.from_numpy(x.reshape(64, nf, pkt).copy()).to(dev).repeat(S // 64, 1, 1).contiguous()
buf = (ctypes.c_ulonglong * 16)()
for k in range(nf):
    if k == nf - 16: f(buf, 1)
    vb.process(d[:, k:k + 1])
f(buf, 0); v = np.array(buf[:9], dtype=np.float64)
names = ['state in', 'packet in (issue)', 'decimation + first split', 'other splits + log energies', 'gaussian probabilities',
         'find_minimum x6', 'model update x6', 'hangover + attenuate + packet out', 'state out (issue)']
waves = S // 64 * 16
for n, c in zip(names, v): print('%-36s %8.0f cycles/wave %5.1f%%' % (n, c / waves, 100 * c / v.sum()))
print('total %.0f cycles/wave' % (v.sum() / waves))
