# developer helper: wall time of every step of bench.py's default chain workload (events around whole steps): where do the
# slow steps sit?  python tools_dev/chain_steps.py [steps]
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
dev = torch.device('cuda:0')
cls, n = bench.WORKLOADS["chain"]
wl = cls(dev, n, 0, None, 1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1300
import gc
if os.environ.get('NO_GC'):
    gc.collect(); gc.disable()
ev = []
for k in range(N):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); wl.step(False); e1.record(); ev.append((e0, e1))
torch.cuda.synchronize()
t = np.array([a.elapsed_time(b) for a, b in ev])
med = np.median(t[300:])
print("median %.4f mean %.4f (steps 300..)" % (med, t[300:].mean()))
print("steps above 1.5x median:", [(int(i), round(float(v), 2)) for i, v in enumerate(t) if v > 1.5 * med][:40])
