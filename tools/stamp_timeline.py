#!/usr/bin/env python3
"""Where does a launch of the multi-asset step kernel spend its time?  (diagnostic; needs the TEMPORARY stamp build)

    python tools/stamp_timeline.py [envs per launch ...]

Needs finenvs_amd/csrc/variants/libfinenvs_amd.stamp.so: the product source with the FE_STAMP part of
profiles/r06_microbench/tile_queue_and_stamps.patch applied and `build_variant('stamp', {'FE_STAMP': 1})` (results:
profiles/r06_microbench/config3_launch_size.md).  Every workgroup of the multi-asset tile loop writes s_memrealtime stamps (100 MHz) into 64 words
of its own: [0] entry, [1] tiles done, then per tile (or per group of FE_STAMP_STRIDE tiles: STAMP_STRIDE=) a pair "accounted" / "streamed".
Back-to-back launches use one stamp buffer each.  Output: phase-1 and streaming time per tile index, how long before the launch's end
the workgroups are done, and the AGGREGATE THROUGHPUT over the launch (every tile's bytes spread over its streaming interval, 100 / 500 us
bins).  STAMP_LIB=tag[,tag ...]: several experiment builds one after the other on ONE shared observation ring."""
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
    tags = os.environ.get("STAMP_LIB", "stamp").split(",")  # several builds: one after the other on ONE shared observation ring
    prices, day_id, _ = make_series(A)
    ring = {}
    for N, tag in [(n, t) for n in sizes for t in tags]:
        print(f"\n# library: variants/libfinenvs_amd.{tag}.so")
        native = _lib.load(os.path.join(os.path.dirname(_lib.LIB_PATH), "variants", f"libfinenvs_amd.{tag}.so"))
        env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1234, obs_buffers=0,
                                        _native=native)
        if N not in ring:
            ring.clear()
            torch.cuda.empty_cache()
            ring[N] = [torch.empty((N, W, 5 * A), dtype=torch.float64, device=DEV) for _ in range(2 if N * W * 5 * A * 8 * 2 < 200e9 else 1)] * 2  # (one buffer twice where two do not fit)
        env._obs_ring, env.obs_buffers = ring[N], 2
        info = env.launch_info()
        grid = info["grid"]
        g = torch.Generator(device=DEV).manual_seed(7)
        acts = [(torch.rand((N, A), generator=g, device=DEV) * 2 - 1).float() for _ in range(2)]
        rew = torch.empty((N,), dtype=torch.float64, device=DEV)
        done = torch.empty((N,), dtype=torch.int32, device=DEV)
        act = torch.empty((N, A), dtype=torch.float32, device=DEV)
        K = 6
        stamps = [torch.zeros((grid, 64), dtype=torch.int64, device=DEV) for _ in range(K)]
        lib, h, st = env._lib, env._handle, torch.cuda.current_stream().cuda_stream
        env.reset()
        for i in range(4):
            env.step(acts[i % 2], rewards_out=rew, dones_out=done, actions_out=act)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        REV = os.environ.get("STAMP_REV") == "1"  # the notify form's walk: tiles from the END of the buffer first (time vs address)
        dev_flag = torch.zeros((1,), dtype=torch.int64, device=DEV)
        for i in range(K):
            _lib.check(lib.fe_env_bind_stats(h, None, None, C.c_void_p(stamps[i].data_ptr())))
            if REV:
                _lib.check(lib.fe_env_step_traj_notify(env._handle_v, acts[i % 2].data_ptr(), env._obs_ring[i % 2].data_ptr(), rew.data_ptr(), done.data_ptr(),
                                                       act.data_ptr(), None, None, C.c_void_p(dev_flag.data_ptr()), i + 1, st))
            else:
                _lib.check(lib.fe_env_step_traj(env._handle_v, acts[i % 2].data_ptr(), env._obs_ring[i % 2].data_ptr(), rew.data_ptr(), done.data_ptr(),
                                                act.data_ptr(), None, None, st))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        B = (40 * W * A + 84 * A + 36) * N
        tile_bytes = (40 * W * A + 84 * A + 36) * info["tile_envs"]
        S = [s.cpu().numpy().astype(np.int64) for s in stamps]
        print(f"\n## {'REVERSED walk (notify form): ' if os.environ.get('STAMP_REV') == '1' else ''}{N} envs x {A} assets x W{W}: grid {grid}, tile {info['tile_envs']} envs, {ms:.3f} ms per launch by HIP events = {B / ms / 1e9 / 8:.3f} of 8 TB/s")
        for i in (3,):
            s, prev = S[i], S[i - 1]
            stride = int(os.environ.get("STAMP_STRIDE", "1"))  # the build's FE_STAMP_STRIDE: one stamp pair per `stride` tiles
            ntiles = s[:, 1]
            nt = np.minimum((ntiles + stride - 1) // stride, 30)
            t0 = s[:, 0].min()
            last = np.array([s[w, 3 + 2 * (nt[w] - 1)] for w in range(grid)])
            plast = np.array([prev[w, 3 + 2 * (min(prev[w, 1], 30) - 1)] for w in range(grid)])
            end = last.max()
            print(f"# launch {i}: previous launch's last workgroup done -> this launch's first workgroup in: {(t0 - plast.max()) * TICK_US:6.1f} us;   "
                  f"first in -> last out: {(end - t0) * TICK_US:9.1f} us")
            print(f"  entry -> first tile accounted (phase 1) (us):        {q((s[:, 2] - s[:, 0]) * TICK_US)}")
            for k in range(0, int(nt.max())):
                has = nt > k
                acc = (s[has, 2 + 2 * k] - (s[has, 1 + 2 * k] if k else s[has, 0])) * TICK_US
                strm = (s[has, 3 + 2 * k] - s[has, 2 + 2 * k]) * TICK_US
                print(f"  tile {k:2d} ({int(has.sum()):4d} workgroups): accounting {np.median(acc):7.1f} us median, {np.percentile(acc, 90):7.1f} p90;   "
                      f"streaming {np.median(strm):7.1f} us median, {np.percentile(strm, 10):7.1f} p10, {np.percentile(strm, 90):7.1f} p90")
            print(f"  workgroup done BEFORE the launch's last one (us):    {q((end - last) * TICK_US)}")
            # aggregate throughput over time: every tile's bytes spread evenly over its streaming interval, 100-us bins
            span = (end - t0) * TICK_US
            binw = 100 if span < 8000 else 500
            nb = int(span // binw) + 1
            bins = np.zeros(nb)
            active = np.zeros(nb)
            for w in range(grid):
                for k in range(nt[w]):
                    a0, a1 = (s[w, 2 + 2 * k] - t0) * TICK_US, (s[w, 3 + 2 * k] - t0) * TICK_US
                    gbytes = tile_bytes * min(stride, int(ntiles[w]) - k * stride)
                    b0, b1 = int(a0 // binw), int(a1 // binw)
                    for bb in range(b0, min(b1, nb - 1) + 1):
                        ov = min(a1, (bb + 1) * binw) - max(a0, bb * binw)
                        if ov > 0:
                            bins[bb] += gbytes * ov / max(a1 - a0, 1e-9)
                            active[bb] += ov / binw
            print(f"  aggregate throughput per {binw} us of the launch (TB/s on B_hbm | workgroups streaming on average):")
            print("   " + "  ".join(f"{bins[j] / (binw * 1e-6) / 1e12:4.2f}|{active[j]:4.0f}" for j in range(nb)))
        lib.fe_env_bind_stats(h, None, None, None)
        env._obs_ring, env.obs_buffers = [], 0
        del env


if __name__ == "__main__":
    main()
