"""GPU box: does splitting the env batch over two HIP streams hide the per-launch start-up / launch boundary?
One env of N envs on one stream vs two envs of N/2 on two streams (envs are independent, so a rollout loop whose policy
is evaluated per half -- a double-buffered sampler -- may do this), each step launched through the C ABI.

    python tools/two_stream_bench.py [config]
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd  # noqa: E402
from bench import CONFIGS, make_series  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
name, N, A, W = CONFIGS[cfg]
prices, day_id, _ = make_series(A)
dev = "cuda:0"
K = 200 if cfg == 2 else 20


def make(n, seed):
    env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=n, redraw="device", seed=seed, obs_buffers=2)
    g = torch.Generator(device=dev).manual_seed(seed)
    acts = [(torch.rand((n, A), generator=g, device=dev) * 2 - 1).float() for _ in range(8)]
    rew = torch.empty((n,), dtype=torch.float64, device=dev)
    done = torch.empty((n,), dtype=torch.int32, device=dev)
    env.reset()
    return env, acts, rew, done


def run(parts, streams):
    """K steps of every part, part p on streams[p]; returns us per step of the whole batch."""
    ts = []
    for rep in range(6):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in streams:
            s.wait_event(e0)
        for k in range(K):
            for (env, acts, rew, done), s in zip(parts, streams):
                env._step_fn(env._handle_v, acts[k % 8].data_ptr(), env._obs_ring[k % 2].data_ptr(), rew.data_ptr(), done.data_ptr(), s.cuda_stream)
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        e1.record()
        torch.cuda.synchronize()
        if rep:
            ts.append(e0.elapsed_time(e1) / K * 1e3)
    return statistics.median(ts)


one = [make(N, 1)]
t1 = run(one, [torch.cuda.Stream()])
del one
torch.cuda.empty_cache()
for parts_n in (2, 4):
    parts = [make(N // parts_n, 10 + i) for i in range(parts_n)]
    t = run(parts, [torch.cuda.Stream() for _ in range(parts_n)])
    print(f"config {cfg}: one env of {N}: {t1:9.2f} us/step; {parts_n} envs of {N // parts_n} on {parts_n} streams: {t:9.2f} us per step of all {N} envs "
          f"({t1 / t:5.3f} x)", flush=True)
    del parts
    torch.cuda.empty_cache()
