#!/usr/bin/env python3
"""Double-buffered sampling: the reference's rollout loop (examples/time_series/PPO_LSTM_training_SPY.py:22-30) over TWO
contiguous shards of the same env batch, each on its own HIP stream.

Envs are independent and the reference's actor acts per env, so `policy(states)` can be evaluated per shard:

    roll = DoubleBufferedRollout(lambda rank, world: TimeSeriesEnv(..., num_envs=N, rank=rank, world_size=world,
                                                                   redraw="device", obs_buffers=2), policy, K)
    for _ in range(replays):
        roll.run()                          # K x (policy -> env.step) per shard, one hipGraph and one HIP stream each
    obs_per_shard = roll.join()             # the shards drift apart between joins: that IS the overlap

The two shards together ARE the unsharded env (same env numbering, same day per env, the evaluation env is the last env of
the second shard; tests/test_sharded_env_gpu.py; this script checks the account state bit for bit), but the GPU now always
has one shard streaming its observations while the other shard's launch boundary, start-up chain (index load -> bar gather
-> accounting) and policy run -- the ~4 us per step that a single 30 us launch cannot hide.  The loops are hipGraphs because
the trick needs a GPU-bound loop: issued eagerly from Python, two shards are twice the host work and the host is the
bottleneck.  Measured at 65 536 envs x W64: pre-generated actions 29.5 -> 26.3 us per step of all envs (C-ABI launches,
tools/two_stream_bench.py; `two_streams` in bench.py's line: 2.4 G env-steps/s, 0.80 of 8 TB/s); with this script's small
torch policy in the graphs 65 -> 57 us (1.15 x).

    python examples/double_buffered_rollout.py [--envs 65536] [--window 64] [--replays 50] [--k 8] [--hidden 32]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from finenvs_amd import TimeSeriesEnv  # noqa: E402
from finenvs_amd.data import synthetic  # noqa: E402
from finenvs_amd.rollout import DoubleBufferedRollout, GraphedRollout  # noqa: E402


class TinyPolicy(torch.nn.Module):
    """A per-env policy (last row of the window -> action), standing in for the reference's actor."""

    def __init__(self, num_obs: int, hidden: int):
        super().__init__()
        self.net = torch.nn.Sequential(torch.nn.Linear(num_obs, hidden), torch.nn.Tanh(), torch.nn.Linear(hidden, 1), torch.nn.Tanh())

    @torch.no_grad()
    def forward(self, states: torch.Tensor) -> torch.Tensor:
        return self.net(states[:, -1, :].float())


def timed(roll, replays):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(replays):
        roll.run()
    if hasattr(roll, "join"):
        roll.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--window", type=int, default=64)
    ap.add_argument("--replays", type=int, default=50)
    ap.add_argument("--k", type=int, default=8, help="steps per hipGraph replay")
    ap.add_argument("--hidden", type=int, default=32)
    args = ap.parse_args()
    prices, day_id, _ = synthetic.synthetic_series(65, 1, 390, 1234)
    kw = dict(prices=prices, day_id=day_id, num_intervals=args.window, num_envs=args.envs, redraw="device", seed=7, obs_buffers=2)
    torch.manual_seed(0)
    policy = TinyPolicy(5, args.hidden).cuda()
    act = lambda states, k: policy(states)  # noqa: E731

    # the loop is captured (finenvs_amd.rollout.GraphedRollout: K iterations of policy -> env.step per replay), so the host
    # is out of the way and what is compared is GPU time
    one = TimeSeriesEnv(**kw)
    roll1 = GraphedRollout(one, act, args.k)
    timed(roll1, 5)
    dt1 = timed(roll1, args.replays)
    cash1 = one.cash.clone()
    del roll1, one
    torch.cuda.empty_cache()

    roll2 = DoubleBufferedRollout(lambda rank, world: TimeSeriesEnv(rank=rank, world_size=world, **kw), act, args.k, shards=2)
    timed(roll2, 5)
    dt2 = timed(roll2, args.replays)
    cash2 = torch.cat([e.cash for e in roll2.envs])

    n = args.envs * args.k * args.replays
    per1, per2 = dt1 / (args.k * args.replays) * 1e6, dt2 / (args.k * args.replays) * 1e6
    print(f"one env of {args.envs}, one graph:          {n / dt1 / 1e9:6.3f} G env-steps/s incl. policy ({per1:6.1f} us per step)")
    print(f"two shards, two graphs, two streams: {n / dt2 / 1e9:6.3f} G env-steps/s incl. policy ({per2:6.1f} us per step of all envs, {dt1 / dt2:5.3f} x)")
    # same envs, same deterministic per-env policy, same number of steps: the account state is the unsharded env's, bit for bit
    print("account state of the two shards == the unsharded env's:", bool(torch.equal(cash1, cash2)))


if __name__ == "__main__":
    main()
