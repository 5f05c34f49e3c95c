#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused TimeSeriesEnv.step() hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W] [--config 2|3|4|1] [--no-cpu]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
one rank per GPU; envs are sharded contiguously (weak scaling: every rank owns the
config's full env count), no collective inside step(); once per TRAJ_T steps the
compact trajectory fields are all-gathered over RCCL.

A "step" is one env.step(actions) over all envs of the workload: synthetic GBM minute
bars (65 business days -> D=64 episodes of 390 bars), a ring of 8
pre-generated uniform action tensors already resident in HBM, training mode, f64
observations (the reference's dtype).  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

CONFIGS = {
    # BASELINE.json configs: name -> (envs per GPU, assets, window)
    1: ("1k envs, 1 asset, window=32", 1024, 1, 32),
    2: ("64k envs, 1 asset, window=64", 65536, 1, 64),
    3: ("256k envs, 30 assets, window=64", 262144, 30, 64),
    4: ("1M envs, 30 assets, window=128", 1048576, 30, 128),
    5: ("512k envs per GPU, 30 assets, window=128 (4M over 8 GPUs)", 524288, 30, 128),
}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
TRAJ_T = 16             # steps per trajectory all-gather (N > 1 only)
# the reference's own PyTorch-CPU path, measured in the build container
REFERENCE_CPU_QUOTED = {"value": 40497, "unit": "env-steps/s", "cores": 8,
                        "what": "hmomin/FinEnvs TimeSeriesEnv.step, torch 2.10 CPU, 65536 envs x W64, build container"}


def algorithmic_bytes(W: int, A: int) -> int:
    """obs write 8*W*5A + window read 8*W*4A + 84 B per sleeve + 36 B per env."""
    return 72 * W * A + 84 * A + 36


def make_series(A: int):
    from finenvs_amd.data import synthetic

    return synthetic.synthetic_series(65, A, 390, 1234)


def cpu_baseline(A: int, W: int, budget_s: float = 12.0):
    """The oracle (C restatement of the reference, OpenMP) on this box's host cores,
    on a bounded sample of the same workload."""
    from oracle import fe_oracle as fo

    fo.build()
    prices, day_id, _ = make_series(A)
    P, LR, *_ = fo.tables_from_series(prices, day_id, W)
    # the GPU box gives one GPU a 16-core share of the host (os.cpu_count() reports the whole machine)
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("FE_CPU_THREADS", "16")))
    n = 65536 if A == 1 else 4096
    g = torch.Generator().manual_seed(7)
    acts = [(torch.rand((n, A), generator=g) * 2 - 1).float().numpy() for _ in range(8)]

    def timed(threads, budget):
        env = fo.OracleEnv(P, LR, W, num_envs=n, redraw_mode=1, seed=1, nthreads=threads)
        env.step(acts[0])
        t0 = time.perf_counter()
        k = 0
        while True:
            env.step(acts[k % 8])
            k += 1
            el = time.perf_counter() - t0
            if el > budget:
                return n * k / el, k, el

    rate, k, el = timed(cores, budget_s)
    rate1, _, _ = timed(1, 3.0)
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next(line.split(":", 1)[1].strip() for line in f if line.startswith("model name"))
    except Exception:  # noqa: BLE001
        pass
    return {"value": rate, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{k} steps of {n} envs x {A} assets x W{W} (oracle/fe_oracle.c, OpenMP {cores} threads, {el:.1f} s)",
            "value_1thread": rate1, "cpu_model": model, "host_cpus_visible": os.cpu_count(),
            "reference_quoted": REFERENCE_CPU_QUOTED}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--redraw", default="device", choices=["device", "torch"])
    ap.add_argument("--obs-f32", action="store_true", help="f32 observations (NOT the reference dtype; extra mode)")
    ap.add_argument("--graph", action="store_true", help="replay the 8-action ring as one hipGraph per 8 steps")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    import torch.distributed as dist

    # rehearsal knobs for a one-GPU box (never set by the driver): all ranks on device 0 + gloo
    if os.environ.get("FE_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("FE_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend)

    import finenvs_amd
    from finenvs_amd.trajectory import TrajectoryBuffer

    name, n_per_gpu, A, W = CONFIGS[args.config]
    prices, day_id, _ = make_series(A)
    # config 4's observation is 153.6 GB: a single env-owned buffer (SURVEY section 7 "Capacity")
    obs_bytes = n_per_gpu * W * 5 * A * (4 if args.obs_f32 else 8)
    obs_buffers = 2 if 2 * obs_bytes < 200e9 else 1
    env = finenvs_amd.TimeSeriesEnv(
        prices=prices, day_id=day_id, num_intervals=W, num_envs=n_per_gpu * world, rank=rank, world_size=world,
        device_id=local_rank, redraw=args.redraw, seed=1234, obs_buffers=obs_buffers,
        obs_dtype=torch.float32 if args.obs_f32 else torch.float64)
    N = env.num_envs
    g = torch.Generator(device=dev).manual_seed(7 + rank)
    actions = [(torch.rand((N, A), generator=g, device=dev) * 2 - 1).float() for _ in range(8)]
    # Compact trajectory fields live in a device buffer; the step kernel writes rewards/dones
    # straight into slot t and the "policy" (the pre-generated action ring) owns the action slots,
    # so storing a step costs nothing extra.  With N > 1 each full chunk is all-gathered (async).
    traj = TrajectoryBuffer(TRAJ_T, N, A, device=dev)
    for chunk in traj._views:  # ring period 8 divides TRAJ_T: slot t always holds ring[t % 8]
        for t in range(TRAJ_T):
            chunk[0][t].copy_(actions[t % 8])

    def one_step(i):
        a, r, d = traj.next_slot()
        obs, rew, done, _ = env.step(a, rewards_out=r, dones_out=d)
        if traj.full():
            if world > 1:
                traj.all_gather_async()  # overlaps the next TRAJ_T steps; waited for before reuse
            else:
                traj.clear()
        return obs

    roll = None
    if args.graph:
        if world > 1 or args.steps % 8:
            sys.exit("--graph: single GPU only, and --steps must be a multiple of 8")
        args.warmup = (args.warmup + 7) // 8 * 8
        from finenvs_amd.rollout import GraphedRollout

        roll = GraphedRollout(env, lambda obs, k: actions[k % 8], 8)
        run_steps = lambda n: [roll.run() for _ in range(n // 8)]
    else:
        env.reset()
        run_steps = lambda n: [one_step(i) for i in range(n)]
    run_steps(args.warmup)

    def fence():
        traj.drain()  # outstanding gathers belong to the timed region
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Kernel duration for the roofline: a second pass of k2 launches issued straight through the C ABI
    # (preallocated outputs, no per-step Python work) so the queue never drains, bracketed by ONE pair of
    # HIP events on the launch stream; the average interval = kernel + the ~1.5 us launch boundary.
    from finenvs_amd import _lib as _fl

    k2 = min(max(args.steps, 20), 400)
    stream = torch.cuda.current_stream().cuda_stream
    obs_b = [t.data_ptr() for t in env._obs_ring]  # same ring as the timed region (keeps the HBM/MALL regime)
    nb = len(obs_b)
    rew_b = torch.empty((N,), dtype=torch.float64, device=dev)
    done_b = torch.empty((N,), dtype=torch.int32, device=dev)
    aptr = [a.data_ptr() for a in actions]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn, h = env._step_fn, env._handle_v
    torch.cuda.synchronize()
    e0.record()
    for i in range(k2):
        rc = fn(h, aptr[i % 8], obs_b[i % nb], rew_b.data_ptr(), done_b.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    _fl.check(rc)
    kern_ms = e0.elapsed_time(e1) / k2

    if rank == 0:
        total_envs = N * world if world == 1 else env.global_num_envs
        B = algorithmic_bytes(W, A) if not args.obs_f32 else algorithmic_bytes(W, A) - 4 * W * 5 * A
        achieved = B * N / (kern_ms * 1e-3) / 1e9
        Bc = B - 8 * W * 4 * A  # window reads are served by L2 / Infinity Cache (tables are <= 64 MB)
        traffic = None
        tpath = os.path.join(REPO, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(f"config{args.config}", {}).get("bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            "metric": "env-steps/sec",
            "value": total_envs * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.obs_f32 else "f64",
            "data": "synthetic",
            "config": {"workload": name, "envs_per_gpu": N, "num_assets": A, "window": W,
                       "obs_buffers": obs_buffers, "eval_redraw": args.redraw,
                       "launch_mode": "hipGraph x8 steps" if args.graph else "eager, one launch per step",
                       "launch": env.launch_info(),
                       "trajectory_slots": TRAJ_T, "trajectory_all_gather_every": TRAJ_T if world > 1 else None,
                       "collective_backend": (backend if world > 1 else None)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "fe_env_kernel (fused step)", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_env_step": B, "units_per_launch": N,
                         # the part of B that cannot come from cache: observation + state + outputs
                         "compulsory_hbm_bytes_per_env_step": Bc,
                         "achieved_compulsory": Bc * N / (kern_ms * 1e-3) / 1e9,
                         "frac_compulsory": Bc * N / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
        }
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(A, W)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
