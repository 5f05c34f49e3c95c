#!/usr/bin/env python3
"""The reference's canonical rollout loop (examples/time_series/PPO_LSTM_training_SPY.py:22-30)
on the MI355X-native env, with a small stand-in LSTM policy instead of the full PPO agent
(agents are plain PyTorch-ROCm user code and out of this repo's scope).

    python examples/time_series_rollout.py [--envs 4096] [--window 4] [--iters 8] [--graph]

eager:  states = env.reset(); loop { actions = policy(states.float()); next, r, d, _ = env.step(actions);
        buffer.store(...); states = next }  -- exactly the reference's loop shape.
--graph: the same K steps captured once into a hipGraph (finenvs_amd.rollout.GraphedRollout).
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from finenvs_amd import TimeSeriesEnv  # noqa: E402
from finenvs_amd.data import synthetic  # noqa: E402
from finenvs_amd.rollout import GraphedRollout  # noqa: E402
from finenvs_amd.stats import EpisodeStats  # noqa: E402
from finenvs_amd.trajectory import TrajectoryBuffer  # noqa: E402


class TinyLSTMPolicy(torch.nn.Module):
    """Shape of the reference's ContinuousActorLSTM (lstm.py:29-57): 1-layer LSTM + linear head."""

    def __init__(self, num_obs: int, hidden: int, num_acts: int):
        super().__init__()
        self.lstm = torch.nn.LSTM(num_obs, hidden, batch_first=True)
        self.head = torch.nn.Linear(hidden, num_acts)

    @torch.no_grad()
    def forward(self, states: torch.Tensor) -> torch.Tensor:
        out, _ = self.lstm(states)
        return torch.tanh(self.head(out[:, -1, :]))


def main(envs=4096, window=4, iters=8, steps=16, graph=False, hidden=64, seed=0):
    torch.manual_seed(seed)
    prices, day_id, _ = synthetic.synthetic_series(12, 1, 390, 1234)
    env = TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=window, num_envs=envs, redraw="device", seed=seed,
                        obs_dtype=torch.float32,  # the agents call states.float() anyway (PPO_agent.py:101)
                        obs_buffers=2)            # opt-in ring: this loop never keeps an observation for more than one step
    args = env.get_env_args()
    policy = TinyLSTMPolicy(args["num_observations"], hidden, args["num_actions"]).to(env.device)
    buffer = TrajectoryBuffer(steps, envs, env.num_assets, device=env.device)
    stats = EpisodeStats(env)
    values = torch.zeros((steps, envs), device=env.device)
    if graph:
        roll = GraphedRollout(env, lambda obs, k: policy(obs), steps, trajectory=buffer)
    else:
        states = env.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(iters):
        if graph:
            states = roll.run()
        else:
            buffer.clear()
            for _ in range(steps):
                actions = policy(states)
                next_states, rewards, dones, _ = env.step(actions)
                buffer.store(actions, rewards, dones)
                states = next_states
        returns, advantages = buffer.returns_and_advantages(values, values[-1], gamma=0.99)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    log = stats.read()
    print(f"{'graph' if graph else 'eager'}: {iters * steps * envs / dt:,.0f} env-steps/s incl. policy; "
          f"episodes finished {log['num_training_episodes']}, mean return {log['mean_training_return']:.4f}, "
          f"returns tensor {tuple(returns.shape)}")
    return log, returns


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--window", type=int, default=4)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--graph", action="store_true")
    a = ap.parse_args()
    main(a.envs, a.window, a.iters, a.steps, a.graph)
