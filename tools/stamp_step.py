"""GPU box: timeline of the single-asset step kernel from the FE_STAMP diagnostic build (four s_memrealtime
stamps per workgroup: start, first tile accounted, first tile streamed, end).  Prints, over the workgroups of
one launch, when they start, how long the first tile's phase 1 takes, when the first tile has streamed and when
they end -- i.e. the start-up ramp and the tail of the launch.  Stamps never touch an output."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd  # noqa: E402
from bench import CONFIGS, make_series  # noqa: E402
from finenvs_amd import _lib  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tag = sys.argv[2] if len(sys.argv) > 2 else "stamp"
name, N, A, W = CONFIGS[cfg]
lib = _lib.load(os.path.join(os.path.dirname(_lib.LIB_PATH), "variants", f"libfinenvs_amd.{tag}.so"))
prices, day_id, _ = make_series(A)
env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1234,
                                obs_buffers=2, _native=lib)
if len(sys.argv) > 4:
    env.set_launch(int(sys.argv[3]), int(sys.argv[4]))
grid = env.launch_info()["grid"]
stamps = torch.zeros((grid, 8), dtype=torch.int64, device="cuda")
dummy = torch.zeros((N,), dtype=torch.float32, device="cuda")
acc = torch.zeros((3,), dtype=torch.float64, device="cuda")
_lib.check(lib.fe_env_bind_stats(env._handle, dummy.data_ptr(), acc.data_ptr(), stamps.data_ptr()), lib)
g = torch.Generator(device="cuda").manual_seed(7)
actions = [(torch.rand((N, A), generator=g, device="cuda") * 2 - 1).float() for _ in range(8)]
env.reset()
rows = []
for i in range(60):
    env.step(actions[i % 8])
    if i >= 40:
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().astype(np.float64) * 0.01  # 100 MHz ticks -> us
        t0 = s[:, 0].min()
        rows.append(np.stack([s[:, 0] - t0, s[:, 1] - s[:, 0], s[:, 2] - t0, s[:, 3] - t0, s[:, 4] - s[:, 0], s[:, 5] - s[:, 4],
                              s[:, 1] - s[:, 5], s[:, 6] - s[:, 6].min(), s[:, 0] - s[:, 6]], axis=1))
r = np.median(np.stack(rows), axis=0)  # per workgroup, median over 20 launches


def q(x):
    return "min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % tuple(np.percentile(x, [0, 10, 50, 90, 100]))


print(f"config {cfg}: {name}; launch {env.launch_info()}; us relative to the launch's first workgroup start (median over 20 launches)")
print("wavefront entry (abs)        ", q(r[:, 7]))
print("   entry -> arguments loaded ", q(r[:, 8]))
print("workgroup start              ", q(r[:, 0]))
print("first tile: start->accounted ", q(r[:, 1]))
print("   index+state+action loads  ", q(r[:, 4]))
print("   bar gather + NaN probe    ", q(r[:, 5]))
print("   accounting + LDS + barrier", q(r[:, 6]))
print("first tile streamed (abs)    ", q(r[:, 2]))
print("workgroup end (abs)          ", q(r[:, 3]))
ends = np.sort(r[:, 3])
print("last workgroup ends at %.2f us; 90 %% have ended by %.2f, 50 %% by %.2f" % (ends[-1], ends[int(0.9 * grid)], ends[grid // 2]))
late = r[:, 7] > 0.5 * r[:, 7].max()
print("late-starting workgroups: %d of %d; by blockIdx %% 8: %s; by blockIdx // (grid/8): %s" % (
    late.sum(), grid, np.bincount(np.arange(grid)[late] % 8, minlength=8).tolist(),
    np.bincount(np.arange(grid)[late] * 8 // grid, minlength=8).tolist()))
lab = np.arange(grid) % 8
print("per XCD label (blockIdx % 8): mean wavefront entry / mean workgroup end (us):")
print("   " + "  ".join(f"{x}: {r[lab == x, 7].mean():5.2f} / {r[lab == x, 3].mean():5.2f}" for x in range(8)))
# absolute time base of consecutive launches: does an XCD that ends early also start the NEXT launch early?
# the same for the LAST of 20 back-to-back launches (no host synchronisation in between)
rows2 = []
for rep in range(10):
    for i in range(20):
        env.step(actions[i % 8])
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.float64) * 0.01
    rows2.append(np.stack([s[:, 6] - s[:, 6].min(), s[:, 3] - s[:, 6].min()], axis=1))
r2 = np.median(np.stack(rows2), axis=0)
print("back-to-back launches, per XCD label: mean wavefront entry / mean workgroup end (us):")
print("   " + "  ".join(f"{x}: {r2[lab == x, 0].mean():5.2f} / {r2[lab == x, 1].mean():5.2f}" for x in range(8)))
