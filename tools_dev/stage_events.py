# developer helper: the individual event times behind bench.py's stage_ms for one workload (python tools_dev/stage_events.py ns_agc_mix_32k)
import sys, runpy, json
sys.path.insert(0, '.')
import bench, torch
wl_name = sys.argv[1]
dev = torch.device('cuda:0')
cls, n = bench.WORKLOADS[wl_name]
wl = cls(dev, int(sys.argv[2]) if len(sys.argv) > 2 else n, 0)
for _ in range(264): wl.step(False)
for _ in range(400): wl.step(True)
for _ in range(16): wl.step("all")
torch.cuda.synchronize()
for k, v in wl.t.ev_all.items():
    print(k, [round(a.elapsed_time(b), 3) for a, b in v])
