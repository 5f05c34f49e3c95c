#!/usr/bin/env python3
"""The reference's evaluation script (examples/time_series/PPO_LSTM_testing_SPY.py:24-54) on the MI355X-native env:
an LSTM actor is run over every trading day of a dataset until each day's episode has ended, and the mean of the
per-day returns is reported -- three ways that give the same returns:

    eager   : the reference's loop verbatim (actor.forward(states.float()) -> env.step, one host read per step)
    graph   : the same loop, K steps per hipGraph replay (GraphedRollout.evaluate_returns)
    fused   : the actor evaluated inside the rollout kernel (FusedLSTMRollout.evaluate_returns)

    python examples/lstm_evaluation.py [--days 64] [--hidden 1024] [--window 4]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from finenvs_amd import TimeSeriesEnv  # noqa: E402
from finenvs_amd.data import synthetic  # noqa: E402
from finenvs_amd.rollout import FusedLSTMRollout, GraphedRollout  # noqa: E402


def main(days=64, hidden=1024, window=4, seed=0, quiet=False):
    torch.manual_seed(seed)
    prices, day_id, _ = synthetic.synthetic_series(days + 1, 1, 390, 1234)  # the first day is the warm-up the reference skips
    lstm = torch.nn.LSTM(5, hidden, num_layers=1, batch_first=True).cuda()  # ContinuousActorLSTM's two modules
    last = torch.nn.Linear(hidden, 1).cuda()
    with torch.no_grad():
        lstm.weight_ih_l0.mul_(20.0)  # an untrained net barely moves: make it trade
        last.weight.mul_(0.5 * hidden ** 0.5)

    @torch.no_grad()
    def actor(states, k=0):
        return torch.tanh(last(lstm(states.float())[0][:, -1, :]))

    def env():  # one env per trading day, evaluate mode: exactly the reference's TimeSeriesEnv(env_name, env_key, 4, evaluate=True)
        return TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=window, evaluate=True, obs_buffers=2)

    results = {}
    e = env()
    states, t0 = e.reset(), time.perf_counter()
    while True:  # the reference's loop
        states, _, _, info = e.step(actor(states))
        if "returns" in info:
            break
    torch.cuda.synchronize()
    results["eager"] = (info["returns"], time.perf_counter() - t0)
    roll = GraphedRollout(env(), actor, 16)
    t0 = time.perf_counter()
    r = roll.evaluate_returns()
    torch.cuda.synchronize()
    results["graph"] = (r, time.perf_counter() - t0)
    fused = FusedLSTMRollout.from_modules(env(), lstm, last)
    t0 = time.perf_counter()
    r = fused.evaluate_returns(chunk=16)
    torch.cuda.synchronize()
    results["fused"] = (r, time.perf_counter() - t0)
    if not quiet:
        for name, (ret, dt) in results.items():
            print(f"{name:6s}: mean day return {ret.mean().item():+.6f} over {ret.numel()} days in {dt * 1e3:7.1f} ms")
    return results


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--days", type=int, default=64)
    ap.add_argument("--hidden", type=int, default=1024, choices=[32, 64, 128, 256, 512, 1024])
    ap.add_argument("--window", type=int, default=4)
    a = ap.parse_args()
    main(a.days, a.hidden, a.window)
