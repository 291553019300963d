# developer helper: what the box's PCIe link gives a pinned host buffer (H2D, D2H, both at once) at the pipeline's transfer size
import time, torch
dev = torch.device('cuda:0')
for mb in (11, 64, 256):
    n = mb << 20
    h_in, h_out = torch.empty(n, dtype=torch.uint8).pin_memory(), torch.empty(n, dtype=torch.uint8).pin_memory()
    d_in, d_out = torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def run(h2d, d2h, reps=20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            if h2d:
                with torch.cuda.stream(s1):
                    d_in.copy_(h_in, non_blocking=True)
            if d2h:
                with torch.cuda.stream(s2):
                    h_out.copy_(d_out, non_blocking=True)
        torch.cuda.synchronize()
        return n * reps / (time.perf_counter() - t0) / 1e9
    run(True, True, 3)
    print(f"{mb} MiB: H2D {run(True, False):.1f} GB/s, D2H {run(False, True):.1f} GB/s, both at once {run(True, True):.1f} GB/s each way")
