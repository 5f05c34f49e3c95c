"""GPU box: does the observation ring's placement matter for the step kernel?  One env, R candidate rings of two
buffers each, K back-to-back launches per block, rings interleaved; prints us/step per ring."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd  # noqa: E402
from bench import CONFIGS, make_series  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
name, N, A, W = CONFIGS[cfg]
prices, day_id, _ = make_series(A)
dev = "cuda:0"
env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1234)
g = torch.Generator(device=dev).manual_seed(7)
actions = [(torch.rand((N, A), generator=g, device=dev) * 2 - 1).float() for _ in range(8)]
rings = [[torch.empty((N, W, 5 * A), dtype=torch.float64, device=dev) for _ in range(2)] for _ in range(R)]
rew = torch.empty((N,), dtype=torch.float64, device=dev)
done = torch.empty((N,), dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
K = 100 if N * W * 5 * A * 8 < 1e9 else 10
times = [[] for _ in range(R)]
for rnd in range(8):
    for i, ring in enumerate(rings):
        ptr = [t.data_ptr() for t in ring]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for k in range(K):
            env._step_fn(env._handle_v, actions[k % 8].data_ptr(), ptr[k % 2], rew.data_ptr(), done.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            times[i].append(e0.elapsed_time(e1) / K * 1e3)
med = [statistics.median(t) for t in times]
for i, m in enumerate(med):
    print(f"ring {i:2d} at {rings[i][0].data_ptr():#x}: {m:9.2f} us/step  ({m / min(med):5.3f} x best)", flush=True)
print(f"config {cfg}: best {min(med):.2f}, worst {max(med):.2f} us/step over {R} rings ({(max(med) / min(med) - 1) * 100:.1f} % spread)")
