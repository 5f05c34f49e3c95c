"""Known-good reference on the same hardware: how fast can this GPU stream plain stores?
(torch fill_ and a d2d copy, sizes matching the bench workloads.)  Run on the GPU box."""
import sys, time, torch
dev = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for gb in (0.168, 1.0, 20.0, 100.0):
    n = int(gb * 1e9 / 8)
    x = torch.empty(n, dtype=torch.float64, device=dev)
    t = timeit(lambda: x.fill_(1.5), 20 if gb < 50 else 5)
    line = f"fill  {gb:7.3f} GB: {gb / t / 1e3:6.2f} TB/s"
    if gb <= 20:
        y = torch.empty_like(x)
        t2 = timeit(lambda: y.copy_(x), 20)
        line += f"   copy (r+w bytes): {2 * gb / t2 / 1e3:6.2f} TB/s"
        del y
    print(line, flush=True)
    del x
    torch.cuda.empty_cache()
