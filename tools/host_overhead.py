"""GPU box: where does the host time of env.step() go when the GPU is not the bottleneck (1k envs)?"""
import cProfile, pstats, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd
from bench import make_series
prices, day_id, _ = make_series(1)
env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=32, num_envs=1024, redraw="device", obs_buffers=1)
a = torch.zeros((1024, 1), device="cuda:0")
env.reset()
for _ in range(1000): env.step(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20000): env.step(a)
torch.cuda.synchronize()
print(f"step(): {(time.perf_counter() - t0) / 20000 * 1e6:.2f} us per call (incl. GPU)")
r = torch.empty(1024, dtype=torch.float64, device="cuda:0"); d = torch.empty(1024, dtype=torch.int32, device="cuda:0")
t0 = time.perf_counter()
for _ in range(20000): env.step(a, rewards_out=r, dones_out=d)
torch.cuda.synchronize()
print(f"step(out=): {(time.perf_counter() - t0) / 20000 * 1e6:.2f} us per call")
fn, h, st = env._step_fn, env._handle_v, torch.cuda.current_stream().cuda_stream
o = env._obs_ring[0].data_ptr(); ap, rp, dp = a.data_ptr(), r.data_ptr(), d.data_ptr()
t0 = time.perf_counter()
for _ in range(20000): fn(h, ap, o, rp, dp, st)
torch.cuda.synchronize()
print(f"raw C-ABI call: {(time.perf_counter() - t0) / 20000 * 1e6:.2f} us per call")
pr = cProfile.Profile(); pr.enable()
for _ in range(20000): env.step(a)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
