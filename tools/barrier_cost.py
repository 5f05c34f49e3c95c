"""GPU box: cost of the fence bench.py puts around a timed block when a process group exists (one rank here)."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
gl = dist.new_group(backend="gloo")
t = torch.zeros(1, device="cuda")
h = torch.zeros(1)


def timeit(name, fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{name:46s} {(time.perf_counter() - t0) / n * 1e6:9.1f} us", flush=True)


timeit("torch.cuda.synchronize()", torch.cuda.synchronize)
timeit("dist.barrier() [nccl]", dist.barrier)
timeit("dist.barrier(device_ids=[0]) [nccl]", lambda: dist.barrier(device_ids=[0]))
timeit("all_reduce(device tensor) + synchronize [nccl]", lambda: (dist.all_reduce(t), torch.cuda.synchronize()))
timeit("dist.barrier(group=gloo)", lambda: dist.barrier(group=gl))
timeit("all_reduce(host tensor, gloo)", lambda: dist.all_reduce(h, group=gl))
dist.destroy_process_group()
