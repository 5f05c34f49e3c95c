"""GPU box: host-side floor of the eager step at small env counts (see tools/README.md).

    python tools/host_floor.py
"""
import os, sys, time, torch, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd
from bench import make_series
prices, day_id, _ = make_series(1)
for N in (1024, 8192, 16384):
    for redraw in ("device", "torch"):
        env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=64, num_envs=N, redraw=redraw, seed=1, obs_buffers=2)
        acts = [(torch.rand((N, 1), device="cuda:0") * 2 - 1).float() for _ in range(8)]
        rew = torch.empty((N,), dtype=torch.float64, device="cuda:0"); done = torch.empty((N,), dtype=torch.int32, device="cuda:0"); ac = torch.empty((N, 1), device="cuda:0")
        for mode in ("plain", "traj"):
            def loop(k):
                if mode == "plain":
                    for i in range(k): env.step(acts[i % 8])
                else:
                    for i in range(k): env.step(acts[i % 8], rewards_out=rew, dones_out=done, actions_out=ac)
            loop(2000); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter(); loop(1000); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 1000 * 1e6)
            print(f"N={N:6d} redraw={redraw:6s} {mode:5s}: {statistics.median(ts):6.2f} us/step")
import cProfile, pstats
env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=64, num_envs=1024, redraw="device", seed=1, obs_buffers=2)
acts = [(torch.rand((1024, 1), device="cuda:0") * 2 - 1).float() for _ in range(8)]
pr = cProfile.Profile(); pr.enable()
for i in range(20000): env.step(acts[i % 8], rewards_out=rew[:1024], dones_out=done[:1024], actions_out=ac[:1024])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
