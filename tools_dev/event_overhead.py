# developer check: what do the per-stage HIP events cost the timed loop of bench.py?
import sys, time; sys.path.insert(0, '.')
import torch, bench
dev = torch.device('cuda:0')
w = bench.ChainWorkload(dev, 65536, 0)
for _ in range(264): w.step(False)
for timed in (False, True, False, True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): w.step(timed)
    torch.cuda.synchronize(); print('events' if timed else 'no events', (time.perf_counter() - t0) / 40 * 1e3, 'ms/step')
