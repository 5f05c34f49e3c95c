"""GPU box: where does a timed block's wall time go when a process group exists? (one rank)"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd  # noqa: E402
from bench import CONFIGS, make_series  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29535")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
name, N, A, W = CONFIGS[2]
prices, day_id, _ = make_series(A)
env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1234, obs_buffers=2)
acts = [(torch.rand((N, A), device="cuda") * 2 - 1).float() for _ in range(8)]
rew = torch.empty((N,), dtype=torch.float64, device="cuda")
done = torch.empty((N,), dtype=torch.int32, device="cuda")
K = 50


def block(kind):
    torch.cuda.synchronize()
    if kind != "none":
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        env.step(acts[i % 8], rewards_out=rew, dones_out=done)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if kind == "barrier":
        dist.barrier()
    elif kind == "allreduce":
        t = torch.zeros(1, device="cuda")
        dist.all_reduce(t)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    return [(b - a) * 1e6 for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4))]


for kind in ("none", "barrier", "allreduce", "none", "barrier"):
    rows = [block(kind) for _ in range(12)][2:]
    med = [sorted(c)[len(c) // 2] for c in zip(*rows)]
    print(f"{kind:10s} issue {med[0]:7.1f}  drain {med[1]:7.1f}  barrier {med[2]:7.1f}  final sync {med[3]:6.1f}  "
          f"-> {(sum(med)) / K:6.2f} us/step", flush=True)
dist.destroy_process_group()
