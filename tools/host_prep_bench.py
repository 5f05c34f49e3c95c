"""GPU box: per-call cost of the fused rollouts at env counts where the host is the bottleneck, for two builds of the
library (e.g. a round-2 variant that calls hipFuncSetAttribute / the occupancy query per launch, and the product, which
caches them).

    python tools/host_prep_bench.py <variant tag | product> [<variant tag | product> ...]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd  # noqa: E402
from finenvs_amd import _lib  # noqa: E402
from finenvs_amd.data import synthetic  # noqa: E402
from finenvs_amd.rollout import FusedLSTMRollout, FusedMLPRollout  # noqa: E402

VAR = os.path.join(os.path.dirname(_lib.LIB_PATH), "variants")
prices, day_id, _ = synthetic.synthetic_series(12, 1, 390, 1234)


def load(tag):
    return _lib.load() if tag == "product" else _lib.load(os.path.join(VAR, f"libfinenvs_amd.{tag}.so"))


def timed(run, reps):
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    host = time.perf_counter() - t0  # host time to ISSUE the calls
    torch.cuda.synchronize()
    return host / reps * 1e6, (time.perf_counter() - t0) / reps * 1e6


def cases(lib):
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(0)
    # the reference's own evaluation shape: 9 envs (SPY dummy days), hidden_dim 1024, W = 4 -> split path, W + 1 launches per step
    env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=4, num_envs=9, redraw="device", obs_buffers=1, _native=lib)
    lstm, lin = torch.nn.LSTM(5, 1024, batch_first=True), torch.nn.Linear(1024, 1)
    roll = FusedLSTMRollout.from_modules(env, lstm, lin)
    yield "lstm split, 9 envs, H=1024, K=1 per call", (lambda: roll.run(1, record_actions=False)), 2000, (env, roll)
    env2 = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=4, num_envs=64, redraw="device", obs_buffers=1, _native=lib)
    lstm2, lin2 = torch.nn.LSTM(5, 128, batch_first=True), torch.nn.Linear(128, 1)
    roll2 = FusedLSTMRollout.from_modules(env2, lstm2, lin2)
    yield "lstm fused, 64 envs, H=128, K=1 per call", (lambda: roll2.run(1, record_actions=False)), 5000, (env2, roll2)
    env3 = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=16, num_envs=64, redraw="device", obs_buffers=1, _native=lib)
    roll3 = FusedMLPRollout(env3, torch.randn((80, 64), generator=g), torch.randn(64, generator=g) * 0.3,
                            torch.randn(64, generator=g) / 8, 0.0)
    yield "mlp fused, 64 envs, H=64, W=16, K=1 per call", (lambda: roll3.run(1, record_actions=False)), 5000, (env3, roll3)


if __name__ == "__main__":
    for tag in sys.argv[1:] or ["product"]:
        lib = load(tag)
        for name, run, reps, keep in cases(lib):
            host, wall = timed(run, reps)
            print(f"{tag:10s} {name:46s} host issue {host:7.2f} us/call   wall {wall:7.2f} us/call", flush=True)
