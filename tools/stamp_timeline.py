#!/usr/bin/env python3
"""Where does a launch of the multi-asset step kernel spend its time?  (diagnostic; needs the TEMPORARY stamp build)

    python tools/stamp_timeline.py [envs per launch ...]

Needs finenvs_amd/csrc/variants/libfinenvs_amd.stamp.so: the product source with the FE_STAMP part of
profiles/r06_microbench/tile_queue_and_stamps.patch applied and `build_variant('stamp', {'FE_STAMP': 1})` (results:
profiles/r06_microbench/config3_launch_size.md).  Every workgroup of the multi-asset tile loop writes s_memrealtime stamps -- 100 MHz --
at entry, after its first accounting, after its first / fifth / last tile, plus HW_ID / XCC_ID).  Back-to-back launches alternate
over two stamp buffers, so the boundary between launch i - 1 and launch i is visible: last workgroup of i - 1 done -> first
workgroup of i in -> ... -> all workgroups streaming."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd  # noqa: E402
from bench import make_series  # noqa: E402
from finenvs_amd import _lib  # noqa: E402

DEV = "cuda:0"
A, W = 30, 128
TICK_US = 0.01  # s_memrealtime: 100 MHz


def q(x):
    x = np.asarray(x, dtype=np.float64)
    return f"min {x.min():9.1f}  p10 {np.percentile(x, 10):9.1f}  median {np.median(x):9.1f}  p90 {np.percentile(x, 90):9.1f}  p99 {np.percentile(x, 99):9.1f}  max {x.max():9.1f}"


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [131072, 262144, 32768]
    native = _lib.load(os.path.join(os.path.dirname(_lib.LIB_PATH), "variants", "libfinenvs_amd.stamp.so"))
    prices, day_id, _ = make_series(A)
    for N in sizes:
        env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1234, obs_buffers=2,
                                        _native=native)
        info = env.launch_info()
        grid = info["grid"]
        g = torch.Generator(device=DEV).manual_seed(7)
        acts = [(torch.rand((N, A), generator=g, device=DEV) * 2 - 1).float() for _ in range(2)]
        rew = torch.empty((N,), dtype=torch.float64, device=DEV)
        done = torch.empty((N,), dtype=torch.int32, device=DEV)
        act = torch.empty((N, A), dtype=torch.float32, device=DEV)
        K = 6
        stamps = [torch.zeros((grid, 8), dtype=torch.int64, device=DEV) for _ in range(K)]
        lib, h, st = env._lib, env._handle, torch.cuda.current_stream().cuda_stream
        env.reset()
        for i in range(4):
            env.step(acts[i % 2], rewards_out=rew, dones_out=done, actions_out=act)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(K):
            _lib.check(lib.fe_env_bind_stats(h, None, None, C.c_void_p(stamps[i].data_ptr())))
            _lib.check(lib.fe_env_step_traj(env._handle_v, acts[i % 2].data_ptr(), env._obs_ring[i % 2].data_ptr(), rew.data_ptr(), done.data_ptr(),
                                            act.data_ptr(), None, None, st))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        B = (40 * W * A + 84 * A + 36) * N
        S = [s.cpu().numpy().astype(np.int64) for s in stamps]
        print(f"\n## {N} envs x {A} assets x W{W}: grid {grid}, tile {info['tile_envs']} envs, {ms:.3f} ms per launch by HIP events = {B / ms / 1e9 / 8:.3f} of 8 TB/s")
        for i in range(2, K):
            s, prev = S[i], S[i - 1]
            t0 = s[:, 0].min()
            prev_end = prev[:, 3].max()
            end = s[:, 3].max()
            tiles = s[:, 4]
            xcc = s[:, 7] & 0xF
            print(f"# launch {i}: previous launch's last workgroup done -> this launch's first workgroup in: {(t0 - prev_end) * TICK_US:8.1f} us;   "
                  f"first in -> last out: {(end - t0) * TICK_US:9.1f} us;   previous last out -> this last out: {(end - prev_end) * TICK_US:9.1f} us")
            print(f"  workgroup entry after the first one (us):            {q((s[:, 0] - t0) * TICK_US)}")
            print(f"  entry -> first tile accounted (phase 1) (us):        {q((s[:, 1] - s[:, 0]) * TICK_US)}")
            print(f"  first tile: accounted -> streamed (us):              {q((s[:, 2] - s[:, 1]) * TICK_US)}")
            has5 = tiles >= 5
            if has5.any():
                print(f"  tiles 2 - 5, per tile (us):                          {q((s[has5, 5] - s[has5, 2]) * TICK_US / 4)}")
                rest = has5 & (tiles > 5)
                if rest.any():
                    print(f"  tiles 6 - last, per tile (us):                       {q((s[rest, 3] - s[rest, 5]) * TICK_US / (tiles[rest] - 5))}")
            print(f"  workgroup done BEFORE the launch's last one (us):    {q((end - s[:, 3]) * TICK_US)}")
            print(f"  tiles per workgroup: {np.bincount(tiles)[tiles.min():].tolist()} from {tiles.min()};   XCC_ID == blockIdx % 8 for {int((xcc == np.arange(grid) % 8).sum())} of {grid} workgroups")
            # how many workgroups are still streaming as the launch ends: the tail
            for back in (400, 200, 100, 50, 20):
                alive = int(((end - s[:, 3]) * TICK_US < back).sum())
                print(f"    finished within the last {back:4d} us: {alive:5d} of {grid} workgroups")
            per_x = [f"xcc {x}: median done {np.median((end - s[xcc == x, 3]) * TICK_US):7.1f} us before the end, entry {np.median((s[xcc == x, 0] - t0) * TICK_US):6.1f} us" for x in sorted(set(xcc.tolist()))]
            print("  " + "\n  ".join(per_x))
            if i >= 3:
                break
        lib.fe_env_bind_stats(h, None, None, None)
        del env
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
