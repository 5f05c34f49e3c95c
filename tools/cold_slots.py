"""GPU box: what trajectory slots that are cold (a new 256 KB / 512 KB / 256 KB region every step) cost the step kernel.

    python tools/cold_slots.py [T]

actions read from / rewards+dones written to: a ring of 8 hot buffers, or T distinct slots of one big chunk."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd  # noqa: E402
from bench import CONFIGS, make_series  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
name, N, A, W = CONFIGS[2]
prices, day_id, _ = make_series(A)
env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1234, obs_buffers=2)
g = torch.Generator(device="cuda").manual_seed(7)
ring = [(torch.rand((N, A), generator=g, device="cuda") * 2 - 1).float() for _ in range(8)]
slots_a = torch.empty((T, N, A), dtype=torch.float32, device="cuda")
for t in range(T):
    slots_a[t].copy_(ring[t % 8])
slots_r = torch.empty((T, N), dtype=torch.float64, device="cuda")
slots_d = torch.empty((T, N), dtype=torch.int32, device="cuda")
hot_r = [torch.empty((N,), dtype=torch.float64, device="cuda") for _ in range(8)]
hot_d = [torch.empty((N,), dtype=torch.int32, device="cuda") for _ in range(8)]


def block(cold_a, cold_rd, refresh):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for t in range(T):
        a = slots_a[t] if cold_a else ring[t % 8]
        if refresh:
            a.copy_(ring[(t + 1) % 8])  # a "policy" writing its actions into the slot just before the step
        env.step(a, rewards_out=slots_r[t] if cold_rd else hot_r[t % 8], dones_out=slots_d[t] if cold_rd else hot_d[t % 8])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / T * 1e3


for cold_a, cold_rd, refresh in ((False, False, False), (True, False, False), (False, True, False), (True, True, False), (True, True, True), (False, False, True)):
    ts = [block(cold_a, cold_rd, refresh) for _ in range(6)][1:]
    print(f"T={T} actions {'slots' if cold_a else 'ring '}  rewards/dones {'slots' if cold_rd else 'ring '}  policy-writes-actions {refresh!s:5}: "
          f"{statistics.median(ts):6.2f} us/step (min {min(ts):.2f})", flush=True)
