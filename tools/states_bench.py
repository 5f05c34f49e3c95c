"""GPU box: what keeping the PPO buffer's `states` as descriptors costs and saves.

    python tools/states_bench.py [config]

(a) env.step with and without descriptors_out (interleaved blocks, one observation ring);
(b) bytes per env-step of the states field: descriptors against the observations the reference's buffer stores;
(c) rendering random minibatches from a T-step trajectory of descriptors (PPO_agent.py:175-188).
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd  # noqa: E402
from bench import CONFIGS, make_series  # noqa: E402
from finenvs_amd.trajectory import TrajectoryBuffer  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
name, N, A, W = CONFIGS[cfg]
T = 32 if cfg == 2 else 8
prices, day_id, _ = make_series(A)
env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1234, obs_buffers=2)
traj = TrajectoryBuffer(T, N, A, states=True)
g = torch.Generator(device="cuda").manual_seed(7)
actions = [(torch.rand((N, A), generator=g, device="cuda") * 2 - 1).float() for _ in range(8)]
env.reset()
traj.begin(env)


def block(described: bool, K: int) -> float:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(K):
        if traj.full():
            traj.clear()
        a, r, d = traj.next_slot()
        env.step(actions[i % 8], rewards_out=r, dones_out=d, descriptors_out=traj.state_slot() if described else None)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3


K = 100 if cfg == 2 else 10
times = {False: [], True: []}
for rnd in range(8):
    for described in (False, True):
        t = block(described, K)
        if rnd:
            times[described].append(t)
base, desc = statistics.median(times[False]), statistics.median(times[True])
print(f"config {cfg}: step {base:.2f} us, with descriptors_out {desc:.2f} us ({(desc / base - 1) * 100:+.2f} %)")
obs_b = W * 5 * A * 8
print(f"states field per env-step: {8 + 8 * A} B as descriptors against {obs_b} B as observations ({obs_b / (8 + 8 * A):.0f}x); "
      f"a {T}-step chunk of {N} envs: {(T + 1) * N * (8 + 8 * A) / 1e6:.1f} MB against {T * N * obs_b / 1e9:.2f} GB")
# (c) fill a whole chunk with described steps, then render minibatches
traj.clear() if traj.full() else None
while not traj.full():
    a, r, d = traj.next_slot()
    a.copy_(actions[len(traj) % 8])
    env.step(a, rewards_out=r, dones_out=d, descriptors_out=traj.state_slot())
B = min(N * T // 4, 1 << 16 if A == 1 else 1 << 13)
idx = torch.randint(0, N * T, (B,), generator=g, device="cuda")
out = None
for _ in range(3):
    out = traj.minibatch_states(env, idx)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    out = traj.minibatch_states(env, idx)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"minibatch of {B} random samples rendered in {ms * 1e3:.1f} us = {B * obs_b / ms / 1e9:.2f} TB/s of observations written "
      f"(index arithmetic + gather of the descriptors included)")
