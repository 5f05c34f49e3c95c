"""GPU box: time reset() (phase 1 = two loads + one division) vs step() (full accounting) on one config."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd
from bench import CONFIGS, make_series
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
name, N, A, W = CONFIGS[cfg]
prices, day_id, _ = make_series(A)
env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device")
g = torch.Generator(device="cuda:0").manual_seed(7)
actions = [(torch.rand((N, A), generator=g, device="cuda:0") * 2 - 1).float() for _ in range(8)]
def timed(fn, n=300):
    for i in range(20): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(3):
    print(f"reset {timed(lambda i: env.reset()):8.2f} us   step {timed(lambda i: env.step(actions[i % 8])):8.2f} us", flush=True)
