"""GPU box: does it matter WHERE the step kernel writes rewards / dones / its copy of the actions?  Trains of back-to-back
fe_env_step_traj launches (as bench.py's kernel_interval_ms) with the three small outputs (a) in the same buffers every
launch, (b) rotating over the T slots of a TrajectoryBuffer, as the timed loop does (T = 2, 16, 64).

    python tools/slot_rotation.py [config]
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd  # noqa: E402
from bench import CONFIGS, make_series  # noqa: E402
from finenvs_amd import _lib  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
name, N, A, W = CONFIGS[cfg]
prices, day_id, _ = make_series(A)
env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1234, obs_buffers=2)
g = torch.Generator(device="cuda:0").manual_seed(7)
actions = [(torch.rand((N, A), generator=g, device="cuda:0") * 2 - 1).float() for _ in range(8)]
env.reset()
stream = torch.cuda.current_stream().cuda_stream
obs_b = [t.data_ptr() for t in env._obs_ring]
fn, h = env._lib.fe_env_step_traj, env._handle_v
aptr = [a.data_ptr() for a in actions]
K = 400


def train(T, with_actions=True):
    rew = torch.empty((T, N), dtype=torch.float64, device="cuda:0")
    done = torch.empty((T, N), dtype=torch.int32, device="cuda:0")
    act = torch.empty((T, N, A), dtype=torch.float32, device="cuda:0")
    rp = [rew[t].data_ptr() for t in range(T)]
    dp = [done[t].data_ptr() for t in range(T)]
    ap = [act[t].data_ptr() if with_actions else None for t in range(T)]
    out = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(K):
            rc = fn(h, aptr[i % 8], obs_b[i % 2], rp[i % T], dp[i % T], ap[i % T], None, None, stream)
        e1.record()
        torch.cuda.synchronize()
        _lib.check(rc)
        if rep:
            out.append(e0.elapsed_time(e1) / K * 1e3)
    return statistics.median(out), min(out), max(out)


print(name)
for rnd in range(2):
    for T in (1, 2, 16, 64, 1):
        med, lo, hi = train(T)
        print(f"  round {rnd}  T = {T:3d} slots: {med:7.2f} us per launch (min {lo:.2f}, max {hi:.2f})")
med, lo, hi = train(16, with_actions=False)
print(f"  T = 16, no action copy: {med:7.2f} us (min {lo:.2f}, max {hi:.2f})")
