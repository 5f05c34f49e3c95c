#!/usr/bin/env python3
"""Where does config 3's 0.70 come from?  One bounded experiment (task 6).

    python tools/slab_ring.py [rounds] [launches per train]

Config 3 (262 144 envs x 30 assets x W64, 20.1 GB per observation) runs at 0.69 - 0.73 of 8 TB/s while the SAME kernel
instantiation reaches 0.80 - 0.85 at config 4 (1 048 576 x 30 x W128, ONE 154 GB buffer).  Rounds 2 - 3 blamed the physical
placement of 20 GB allocations.  This script separates the candidates in one process, with one set of envs, interleaved trains of
back-to-back C-ABI launches (fe_env_step_traj, lean form, redraw='device'), HIP events per train, median over rounds:

  placement   the config-3 env writing (i) its ring of two separately allocated 20 GB buffers, (ii) two 20 GB carves of one 60 GB
              slab, (iii) 20 GB carves of a 154 GB slab (the config-4 buffer) at several offsets
  alternation the same env writing ONE buffer every launch vs alternating over two
  window      an env of the same bytes per launch with W = 128 (131 072 envs x 30 x W128 = 20.1 GB) on the SAME buffers as (i) / (iii),
              and the config-4 env itself on the whole slab

Writes a table to stdout (copy it to profiles/r06_microbench/config3_launch_size.md)."""
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
GB = 1 << 30


class Arm:
    def __init__(self, name, env, bufs, actions):
        self.name, self.env, self.bufs, self.actions = name, env, bufs, actions
        N, A = env.num_envs, env.num_assets
        self.rew = torch.empty((N,), dtype=torch.float64, device=DEV)
        self.done = torch.empty((N,), dtype=torch.int32, device=DEV)
        self.act = torch.empty((N, A), dtype=torch.float32, device=DEV)
        self.bytes = (40 * env.num_intervals * A + 84 * A + 36) * N  # B_hbm per launch (bench.hbm_bytes)
        self.ms = []

    def train(self, k):
        env, fn = self.env, self.env._lib.fe_env_step_traj
        st = torch.cuda.current_stream().cuda_stream
        h, rp, dp, ap = env._handle_v, self.rew.data_ptr(), self.done.data_ptr(), self.act.data_ptr()
        a = [t.data_ptr() for t in self.actions]
        nb = len(self.bufs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rc = 0
        e0.record()
        for i in range(k):
            rc = fn(h, a[i % len(a)], self.bufs[i % nb], rp, dp, ap, None, None, st) or rc
        e1.record()
        torch.cuda.synchronize()
        if rc:
            _fl.check(rc)
        return e0.elapsed_time(e1) / k


def carve(slab, offset, nbytes):
    assert offset % (2 << 20) == 0 and offset + nbytes <= slab.numel()
    return slab.data_ptr() + offset


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    t00 = time.perf_counter()
    prices, day_id, _ = make_series(30)

    def mk(N, W):
        env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1234, obs_buffers=0)
        g = torch.Generator(device=DEV).manual_seed(7)
        acts = [(torch.rand((N, 30), generator=g, device=DEV) * 2 - 1).float() for _ in range(4)]
        return env, acts

    c3, a3 = mk(262144, 64)      # config 3
    w128, a128 = mk(131072, 128)  # the same bytes per launch at W = 128
    nbytes = 262144 * 64 * 150 * 8
    assert nbytes == 131072 * 128 * 150 * 8
    print(f"# observation of config 3: {nbytes / 1e9:.2f} GB per launch; B_hbm per launch {(40 * 64 * 30 + 84 * 30 + 36) * 262144 / 1e9:.2f} GB "
          f"(W64) / {(40 * 128 * 30 + 84 * 30 + 36) * 131072 / 1e9:.2f} GB (W128)", flush=True)
    ring = [torch.empty(nbytes, dtype=torch.uint8, device=DEV) for _ in range(2)]
    slab60 = torch.empty(60 * GB, dtype=torch.uint8, device=DEV)
    step20 = (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    arms = [
        Arm("c3 W64: ring of two separate 20 GB allocations (today)", c3, [t.data_ptr() for t in ring], a3),
        Arm("c3 W64: ring member 0 only (no alternation)", c3, [ring[0].data_ptr()], a3),
        Arm("c3 W64: ring member 1 only (no alternation)", c3, [ring[1].data_ptr()], a3),
        Arm("c3 W64: two 20 GB carves of one 60 GB slab", c3, [carve(slab60, 0, nbytes), carve(slab60, step20, nbytes)], a3),
        Arm("c3 W64: 60 GB slab, carve 0 only", c3, [carve(slab60, 0, nbytes)], a3),
        Arm("W128 x 131072 envs: ring of two separate 20 GB allocations (same buffers)", w128, [t.data_ptr() for t in ring], a128),
        Arm("W128 x 131072 envs: ring member 0 only", w128, [ring[0].data_ptr()], a128),
        Arm("W128 x 131072 envs: two carves of the 60 GB slab", w128, [carve(slab60, 0, nbytes), carve(slab60, step20, nbytes)], a128),
    ]

    def run(arms, label):
        for arm in arms:  # first touch + translations
            arm.train(2 * len(arm.bufs))
        for r in range(rounds):
            for arm in arms:
                arm.train(4)  # settle on this arm's buffers
                arm.ms.append(arm.train(k))
            print(f"# {label}: round {r + 1}/{rounds} done at {time.perf_counter() - t00:.0f} s", flush=True)

    run(arms, "phase 1 (ring, 60 GB slab)")
    done_arms = list(arms)
    del slab60
    for arm in arms:
        if "slab" in arm.name:
            arm.bufs = []
    torch.cuda.empty_cache()
    time.sleep(2.0)
    free = torch.cuda.mem_get_info(DEV)[0]
    print(f"# free before the 154 GB slab: {free / GB:.1f} GiB", flush=True)
    c4_bytes = 1048576 * 128 * 150 * 8
    if free > c4_bytes + 8 * GB:
        slab154 = torch.empty(c4_bytes, dtype=torch.uint8, device=DEV)
        c4, a4 = mk(1048576, 128)
        offs = [0, step20, 3 * step20, 6 * step20]
        arms2 = [
            Arm("c3 W64: ring of two separate 20 GB allocations (again, beside phase 2)", c3, [t.data_ptr() for t in ring], a3),
            Arm("c3 W64: carves 0 and 1 of the 154 GB slab", c3, [carve(slab154, offs[0], nbytes), carve(slab154, offs[1], nbytes)], a3),
            Arm("c3 W64: carves 3 and 6 of the 154 GB slab", c3, [carve(slab154, offs[2], nbytes), carve(slab154, offs[3], nbytes)], a3),
            Arm("c3 W64: carve 0 of the 154 GB slab only", c3, [carve(slab154, offs[0], nbytes)], a3),
            Arm("W128 x 131072 envs: carves 0 and 1 of the 154 GB slab", w128, [carve(slab154, offs[0], nbytes), carve(slab154, offs[1], nbytes)], a128),
            Arm("config 4 itself: 1 048 576 envs x W128 on the whole 154 GB slab", c4, [slab154.data_ptr()], a4),
        ]
        k_keep = k
        for arm in arms2:
            arm.train(2 * len(arm.bufs))
        for r in range(rounds):
            for arm in arms2:
                kk = max(4, k_keep // 6) if arm.env is c4 else k_keep
                arm.train(2 if arm.env is c4 else 4)
                arm.ms.append(arm.train(kk))
            print(f"# phase 2 (154 GB slab): round {r + 1}/{rounds} done at {time.perf_counter() - t00:.0f} s", flush=True)
        done_arms += arms2
    else:
        print("# not enough free memory for the 154 GB slab: phase 2 skipped", flush=True)
    print(f"\n| arm | ms per launch (median of {rounds}) | min | max | TB/s on B_hbm | frac of 8 TB/s |\n|---|---|---|---|---|---|")
    for arm in done_arms:
        med = statistics.median(arm.ms)
        print(f"| {arm.name} | {med:.3f} | {min(arm.ms):.3f} | {max(arm.ms):.3f} | {arm.bytes / med / 1e9:.2f} | {arm.bytes / med / 1e9 / 8.0:.3f} |")
    print(f"\n# {rounds} rounds of {k}-launch trains, interleaved arm by arm; total {time.perf_counter() - t00:.0f} s")


if __name__ == "__main__":
    main()
