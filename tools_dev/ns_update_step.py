# developer helper: per-launch time of the noise suppressor around its 500-block threshold updates (all streams of the batch
# reach the update in the same launch here; in a deployment the streams are staggered)
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from wmix_amd import synth
from wmix_amd.ns import NsBatch
dev = torch.device('cuda:0'); S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nb = NsBatch(S, 1, 16000)
K = 200
x = synth.ns_input(7, 256, K, 160)
d = torch.from_numpy(x.reshape(256, K, 160).copy()).to(dev)
idx = torch.arange(S, device=dev) % 256
inp = d[idx].transpose(0, 1).contiguous()  # [K, S, 160]
work = torch.empty_like(inp[0:1])
ev = []
for k in range(1100):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); nb.process_packet_major(inp[k % K:k % K + 1], work); e1.record(); ev.append((e0, e1))
torch.cuda.synchronize()
t = np.array([a.elapsed_time(b) for a, b in ev])
print("median %.3f ms; launches above 2x median:" % np.median(t), [(int(i), round(float(v), 2)) for i, v in enumerate(t) if v > 2 * np.median(t)])
