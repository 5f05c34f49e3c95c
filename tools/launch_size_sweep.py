#!/usr/bin/env python3
"""Multi-asset step kernel: time per launch against bytes per launch, placement held equal.

    python tools/launch_size_sweep.py [covers]

tools/slab_ring.py (profiles/r06_microbench/config3_launch_size.md) showed that 20 GB launches run at 0.69 - 0.76 of 8 TB/s on ANY 20 GB
piece of a 154 GB slab while ONE launch over the whole slab runs at 0.825: not placement, not the window.  This sweep holds the
memory equal -- every arm writes the WHOLE slab, k launches of 1/k of it each, round robin -- and varies only the launch size
(envs per launch, 30 assets, W = 128): if t(launch) = t0 + bytes / BW, the intercept t0 is a per-launch cost inside the kernel and
the slope the bandwidth the kernel reaches once it is running.  Second table: the 131 072-env launch under other grids / tile sizes
(env.set_launch), to see what t0 is made of."""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd  # noqa: E402
from bench import make_series  # noqa: E402
from finenvs_amd import _lib as _fl  # noqa: E402

DEV = "cuda:0"
A, W = 30, 128
FULL = 1048576


def main():
    covers = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    t00 = time.perf_counter()
    prices, day_id, _ = make_series(A)
    per_env = W * 5 * A * 8
    slab = torch.empty(FULL * per_env, dtype=torch.uint8, device=DEV)
    base = slab.data_ptr()
    g = torch.Generator(device=DEV).manual_seed(7)
    acts_full = [(torch.rand((FULL, A), generator=g, device=DEV) * 2 - 1).float() for _ in range(2)]

    variant = os.environ.get("SWEEP_LIB")  # an experiment build (finenvs_amd.csrc.build.build_variant), by tag
    native = _fl.load(os.path.join(os.path.dirname(_fl.LIB_PATH), "variants", f"libfinenvs_amd.{variant}.so")) if variant else None
    print(f"# library: {variant or 'product'}", flush=True)

    def mk(N):
        env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1234, obs_buffers=0,
                                        _native=native)
        rew = torch.empty((N,), dtype=torch.float64, device=DEV)
        done = torch.empty((N,), dtype=torch.int32, device=DEV)
        act = torch.empty((N, A), dtype=torch.float32, device=DEV)
        return env, rew, done, act

    def train(envrec, k_parts, launches):
        """`launches` launches, launch i writing part i % k_parts of the slab (and reading the matching slice of the action tensors)."""
        env, rew, done, act = envrec
        N = env.num_envs
        fn, h, st = env._lib.fe_env_step_traj, env._handle_v, torch.cuda.current_stream().cuda_stream
        rp, dp, ap = rew.data_ptr(), done.data_ptr(), act.data_ptr()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rc = 0
        e0.record()
        for i in range(launches):
            part = i % k_parts
            a = acts_full[(i // k_parts) % 2].data_ptr() + part * N * A * 4
            rc = fn(h, a, base + part * N * per_env, rp, dp, ap, None, None, st) or rc
        e1.record()
        torch.cuda.synchronize()
        if rc:
            _fl.check(rc)
        return e0.elapsed_time(e1) / launches

    rows = []
    envs = {}
    for k in (1, 2, 4, 8, 16, 32, 64):
        envs[k] = mk(FULL // k)
        train(envs[k], k, k)  # first touch of the slab / translations
    print(f"# built at {time.perf_counter() - t00:.0f} s", flush=True)
    for rnd in range(3):
        for k in (1, 2, 4, 8, 16, 32, 64):
            launches = max(k * covers, 8)
            train(envs[k], k, max(k, 2))
            rows.append((k, train(envs[k], k, launches)))
        print(f"# round {rnd + 1}/3 at {time.perf_counter() - t00:.0f} s", flush=True)
    print("\n| envs per launch | launches per cover of the slab | GB (B_hbm) per launch | ms per launch (median of 3) | TB/s | frac of 8 TB/s | ms per WHOLE slab |\n|---|---|---|---|---|---|---|")
    pts = []
    for k in (1, 2, 4, 8, 16, 32, 64):
        N = FULL // k
        ms = statistics.median(t for kk, t in rows if kk == k)
        B = (40 * W * A + 84 * A + 36) * N
        pts.append((B / 1e9, ms))
        print(f"| {N} | {k} | {B / 1e9:.2f} | {ms:.3f} | {B / ms / 1e9:.2f} | {B / ms / 1e9 / 8:.3f} | {ms * k:.2f} |")
    # least squares t = t0 + GB / BW
    n = len(pts)
    sx, sy = sum(p[0] for p in pts), sum(p[1] for p in pts)
    sxx, sxy = sum(p[0] * p[0] for p in pts), sum(p[0] * p[1] for p in pts)
    slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    t0 = (sy - slope * sx) / n
    print(f"\n# fit t = t0 + bytes / BW over the seven sizes: t0 = {t0 * 1e3:.0f} us, BW = {1 / slope:.2f} TB/s ({1 / slope / 8:.3f} of 8 TB/s)")

    # ---- what t0 is made of: the 131 072-env launch (k = 8) under other launch geometries
    env, rew, done, act = envs[8]
    info0 = env.launch_info()
    print(f"\n# 131 072 envs per launch, automatic geometry: {info0}")
    print("| tile_envs | grid | ms per launch | frac of 8 TB/s |\n|---|---|---|---|")
    B = (40 * W * A + 84 * A + 36) * (FULL // 8)
    for tile, grid in ((0, 0), (0, 1024), (0, 512), (0, 256), (4, 0), (1, 0), (2, 2048), (8, 3072), (8, 4096), (8, 8192), (8, 16384), (4, 32768), (2, 65536), (1, 131072), (4, 8192), (2, 8192), (0, 0)):
        try:
            info = env.set_launch(tile_envs=tile, grid=grid)
        except Exception as exc:  # noqa: BLE001
            print(f"| {tile} | {grid} | refused: {exc} | |")
            continue
        train(envs[8], 8, 8)
        ms = statistics.median(train(envs[8], 8, 16) for _ in range(3))
        print(f"| {info['tile_envs']} | {info['grid']} | {ms:.3f} | {B / ms / 1e9 / 8:.3f} |", flush=True)
    env.set_launch(0, 0)

    # ---- does overlapping consecutive launches recover it?  The same parts of the slab, alternately on two / three / four streams
    # (two env objects per size: a launch owns its env's state)
    print("\n| envs per launch | streams | ms per WHOLE slab | frac of 8 TB/s |\n|---|---|---|---|")
    Bfull = (40 * W * A + 84 * A + 36) * FULL
    for k in (8, 16, 4):
        N = FULL // k
        for ns in (1, 2, 3, 4):
            recs = [envs[k]] + [mk(N) for _ in range(ns - 1)]
            streams = [torch.cuda.Stream() for _ in range(ns)]

            def cover(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for s_ in streams:
                    s_.wait_event(e0)
                rc = 0
                for i in range(reps * k):
                    part = i % k
                    env, rew, done, act = recs[i % ns]
                    a = acts_full[(i // k) % 2].data_ptr() + part * N * A * 4
                    rc = env._lib.fe_env_step_traj(env._handle_v, a, base + part * N * per_env, rew.data_ptr(), done.data_ptr(), act.data_ptr(),
                                                   None, None, streams[i % ns].cuda_stream) or rc
                for s_ in streams:
                    torch.cuda.current_stream().wait_stream(s_)
                e1.record()
                torch.cuda.synchronize()
                if rc:
                    _fl.check(rc)
                return e0.elapsed_time(e1) / reps

            cover(1)
            ms = statistics.median(cover(covers) for _ in range(3))
            print(f"| {N} | {ns} | {ms:.2f} | {Bfull / ms / 1e9 / 8:.3f} |", flush=True)
            del recs
    # ---- control: is it this kernel at all?  torch's fill kernel over the same parts of the same slab
    print("\n| control: slab part filled by torch (uint8 view as int64 .fill_) | parts per cover | ms per WHOLE slab | TB/s written |\n|---|---|---|---|")
    words = slab.view(torch.int64)
    nw = words.numel()
    for k in (1, 2, 4, 8, 16, 32, 64):
        step_w = nw // k
        views = [words[i * step_w:(i + 1) * step_w] for i in range(k)]

        def cover_fill(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                for v in views:
                    v.fill_(7)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        cover_fill(1)
        ms = statistics.median(cover_fill(covers) for _ in range(3))
        print(f"| {step_w * 8 / 1e9:.2f} GB per launch | {k} | {ms:.2f} | {nw * 8 / ms / 1e9:.2f} |", flush=True)
    # ---- where in the slab?  The same 16 parts, each timed on its own: this kernel (65 536 envs per launch, 6 launches back to back
    # into ONE part) and torch's fill of the same bytes.  If both see the same slow / fast parts it is the memory; if only this
    # kernel does, it is the kernel's access pattern on that memory.
    print("\n| part of the slab (1/16 = 10 GB each) | this kernel: TB/s on B_hbm | torch fill: TB/s |\n|---|---|---|")
    k = 16
    N = FULL // k
    env, rew, done, act = envs[k]
    fn, h, st = env._lib.fe_env_step_traj, env._handle_v, torch.cuda.current_stream().cuda_stream
    Bp = (40 * W * A + 84 * A + 36) * N
    step_w = nw // k
    for part in range(k):
        def mine(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = 0
            for i in range(reps):
                rc = fn(h, acts_full[i % 2].data_ptr() + part * N * A * 4, base + part * N * per_env, rew.data_ptr(), done.data_ptr(), act.data_ptr(),
                        None, None, st) or rc
            e1.record()
            torch.cuda.synchronize()
            if rc:
                _fl.check(rc)
            return e0.elapsed_time(e1) / reps

        def fill(reps):
            v = words[part * step_w:(part + 1) * step_w]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                v.fill_(7)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        mine(2)
        m = statistics.median(mine(6) for _ in range(3))
        fill(2)
        f = statistics.median(fill(6) for _ in range(3))
        print(f"| {part} | {Bp / m / 1e9:.2f} | {step_w * 8 / f / 1e9:.2f} |", flush=True)
    print(f"\n# total {time.perf_counter() - t00:.0f} s")


if __name__ == "__main__":
    main()
