"""GPU box: the reference's evaluation loop (PPO_LSTM_testing_SPY.py:43-52) with the torch nn.LSTM actor at small env counts
(N = trading days), three ways: eager step by step, K steps per hipGraph replay, and -- for comparison -- the fused kernel.

    python tools/eval_loop_bench.py [N] [H]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd  # noqa: E402
from finenvs_amd.data import synthetic  # noqa: E402
from finenvs_amd.rollout import FusedLSTMRollout, GraphedRollout  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
W, K = 4, 16
prices, day_id, _ = synthetic.synthetic_series(12, 1, 390, 1234)
torch.manual_seed(0)
lstm, lin = torch.nn.LSTM(5, H, batch_first=True).cuda(), torch.nn.Linear(H, 1).cuda()


@torch.no_grad()
def actor(states, k=0):
    return torch.tanh(lin(lstm(states.float())[0][:, -1, :]))


def mk():
    return finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, evaluate=True, obs_buffers=2)


def timed(make):
    """`make()` builds env + rollout object once (not timed) and returns the loop; the loop runs one whole evaluation (until
    every env has finished an episode) and returns its step count.  First evaluation warms up, the second is timed."""
    loop = make()
    loop()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = loop()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


def eager():
    env = mk()
    box = [env.reset()]

    def loop():
        for t in range(1, 5000):
            box[0], _, _, info = env.step(actor(box[0]))
            if "returns" in info:
                return t
    return loop


def graphed():
    env = mk()
    roll = GraphedRollout(env, actor, K)

    def loop():
        steps = 0
        while True:
            roll.run()
            steps += K
            if "returns" in roll.info:
                return steps
    return loop


def fused():
    env = mk()
    roll = FusedLSTMRollout.from_modules(env, lstm, lin)

    def loop():
        steps = 0
        while True:
            roll.run(K, record_actions=False)
            steps += K
            if int(env._counters[0].item()) == env.num_envs:
                env.reset_evaluation_metrics()
                return steps
    return loop


for name, fn in (("eager, one host read per step", eager), (f"hipGraph, {K} steps per replay", graphed), (f"fused kernel, {K} steps per launch", fused)):
    print(f"N={N} H={H} W={W} evaluation loop, {name:36s}: {timed(fn):9.1f} us/step", flush=True)
