#!/usr/bin/env python3
"""GPU box: when do the workgroups of the SINGLE-asset step kernel finish, per XCD?  (config 2; needs the stamp build of
profiles/r06_microbench/tile_queue_and_stamps.patch, part 5: build_variant('stamp1', {'FE_STAMP': 1}))

The multi-asset launches end with the slower half of the XCDs still working (profiles/r06_microbench/config3_launch_size.md, table 12).  The single-asset
table fits every XCD's L2, so an uneven split would be free there -- but there is nothing to split: at config 2 the eight XCDs' median workgroups finish within
0.7 us of each other (26.7 - 27.4 us of a 29 us launch under the stamps), and the launch's own tail (median -> last workgroup) is 2.2 us."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import finenvs_amd
from bench import make_series
from finenvs_amd import _lib
native = _lib.load(os.path.join(os.path.dirname(_lib.LIB_PATH), "variants", "libfinenvs_amd.stamp1.so"))
prices, day_id, _ = make_series(1)
N, W = 65536, 64
env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1, obs_buffers=2, _native=native)
grid = env.launch_info()["grid"]
g = torch.Generator(device="cuda:0").manual_seed(7)
acts = [(torch.rand((N, 1), generator=g, device="cuda:0") * 2 - 1).float() for _ in range(8)]
rew = torch.empty((N,), dtype=torch.float64, device="cuda:0"); done = torch.empty((N,), dtype=torch.int32, device="cuda:0"); act = torch.empty((N, 1), device="cuda:0")
K = 40
stamps = [torch.zeros((grid, 4), dtype=torch.int64, device="cuda:0") for _ in range(K)]
lib, h, st = env._lib, env._handle, torch.cuda.current_stream().cuda_stream
env.reset()
for i in range(200): env.step(acts[i % 8], rewards_out=rew, dones_out=done, actions_out=act)
torch.cuda.synchronize()
for i in range(K):
    _lib.check(lib.fe_env_bind_stats(h, None, None, C.c_void_p(stamps[i].data_ptr())))
    _lib.check(lib.fe_env_step_traj(env._handle_v, acts[i % 8].data_ptr(), env._obs_ring[i % 2].data_ptr(), rew.data_ptr(), done.data_ptr(), act.data_ptr(), None, None, st))
torch.cuda.synchronize()
S = [s.cpu().numpy().astype(np.int64) for s in stamps]
print(f"config 2, grid {grid}: per-XCD finishing times over launches 10..{K-1} (us, s_memrealtime 100 MHz)")
rows = []
for i in range(10, K):
    s = S[i]; t0 = s[:, 0].min(); end = s[:, 1].max(); x = s[:, 2] & 0xF
    rows.append([np.median((s[x == k, 1] - t0) * 0.01) for k in range(8)] + [(end - t0) * 0.01, (S[i][:, 0].min() - S[i - 1][:, 1].max()) * 0.01])
rows = np.array(rows)
print("median over launches of [median workgroup finish per XCC 0..7 | last workgroup out | gap from the previous launch's last out to this first in]:")
print(np.round(np.median(rows, axis=0), 2))
print("workgroups per XCC:", np.bincount(S[-1][:, 2] & 0xF).tolist(), " XCC == blockIdx % 8:", int(((S[-1][:, 2] & 0xF) == np.arange(grid) % 8).sum()), "of", grid)
s = S[-1]; t0 = s[:, 0].min()
fin = (s[:, 1] - t0) * 0.01
print("finish time of a launch's workgroups: min %.2f p10 %.2f median %.2f p90 %.2f max %.2f us" % (fin.min(), np.percentile(fin, 10), np.median(fin), np.percentile(fin, 90), fin.max()))
