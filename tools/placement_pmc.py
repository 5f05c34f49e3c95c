"""GPU box, UNDER rocprofv3 --pmc ...: which memory-side counters separate a slow 20 GB buffer from a fast one?
Allocates 10 config-3-sized buffers, times the render kernel into each, then -- as the LAST launches of the process --
renders 4 x into the slowest and 4 x into the fastest (tools/placement_pmc_box.sh reads those rows of the counter CSV)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd  # noqa: E402
from bench import CONFIGS, make_series  # noqa: E402
from finenvs_amd import _lib  # noqa: E402

name, N, A, W = CONFIGS[3]
prices, day_id, _ = make_series(A)
env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1234)
n = N * W * 5 * A
st = torch.cuda.current_stream().cuda_stream
bufs = [torch.empty((n,), dtype=torch.float64, device="cuda") for _ in range(10)]


def t_into(p):
    ts = []
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(env._lib.fe_env_reset_obs(env._handle, p, st))
        e1.record()
        e1.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)


times = [t_into(b.data_ptr()) for b in bufs]
slow, fast = max(range(10), key=lambda i: times[i]), min(range(10), key=lambda i: times[i])
print("times ms (under the profiler):", [round(t, 3) for t in times], "slow", slow, "fast", fast, flush=True)
torch.cuda.synchronize()
for i in (slow, fast):
    for _ in range(4):
        _lib.check(env._lib.fe_env_reset_obs(env._handle, bufs[i].data_ptr(), st))
    torch.cuda.synchronize()
