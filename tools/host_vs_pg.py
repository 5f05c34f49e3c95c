"""GPU box: does an initialised RCCL process group slow the host side of the step loop down?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd  # noqa: E402
from bench import CONFIGS, make_series  # noqa: E402

name, N, A, W = CONFIGS[2]
prices, day_id, _ = make_series(A)
env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1234, obs_buffers=2)
acts = [(torch.rand((N, A), device="cuda") * 2 - 1).float() for _ in range(8)]
rew = torch.empty((N,), dtype=torch.float64, device="cuda")
done = torch.empty((N,), dtype=torch.int32, device="cuda")


def measure(tag):
    for _ in range(50):
        env.step(acts[0], rewards_out=rew, dones_out=done)
    torch.cuda.synchronize()
    K = 400
    t0 = time.perf_counter()
    for i in range(K):
        env.step(acts[i % 8], rewards_out=rew, dones_out=done)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{tag:34s} host issue {t_host / K * 1e6:6.1f} us/step   wall {t_all / K * 1e6:6.1f} us/step", flush=True)


measure("no process group")
import torch.distributed as dist  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
measure("after init_process_group (lazy)")
t = torch.zeros(1, device="cuda")
dist.all_reduce(t)
torch.cuda.synchronize()
measure("after the first collective")
x = torch.zeros(1 << 20, device="cuda")
out = torch.empty_like(x)
w = dist.all_gather_into_tensor(out, x, async_op=True)
w.wait()
torch.cuda.synchronize()
measure("after an async all_gather")
print("threads:", len(os.listdir(f"/proc/{os.getpid()}/task")), "cpus:", len(os.sched_getaffinity(0)))
dist.destroy_process_group()
measure("after destroy_process_group")
