"""GPU box: step one build ("product" or an experiment build by name) K times at a bench config -- the thing to
put under rocprofv3 when two builds have to be profiled on the SAME box (tools/ab_prof_box.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd  # noqa: E402
from bench import CONFIGS, make_series  # noqa: E402
from finenvs_amd import _lib  # noqa: E402

cfg, arm = int(sys.argv[1]), sys.argv[2]
K = int(sys.argv[3]) if len(sys.argv) > 3 else 400
name, N, A, W = CONFIGS[cfg]
lib = _lib.load() if arm == "product" else _lib.load(os.path.join(os.path.dirname(_lib.LIB_PATH), "variants", f"libfinenvs_amd.{arm}.so"))
prices, day_id, _ = make_series(A)
obs_bytes = N * W * 5 * A * 8
env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1234,
                                obs_buffers=2 if 2 * obs_bytes < 200e9 else 1, _native=lib)
g = torch.Generator(device="cuda").manual_seed(7)
actions = [(torch.rand((N, A), generator=g, device="cuda") * 2 - 1).float() for _ in range(8)]
env.reset()
for i in range(20 + K):
    env.step(actions[i % 8])
torch.cuda.synchronize()
print(f"{arm}: {K} steps of config {cfg} done ({lib.fe_build_tag().decode() or 'product build'})")
