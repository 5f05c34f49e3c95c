"""Extra legs of bench.py (never part of `value`): the same workload family under other launch modes.

bench.py imports this module lazily AFTER the headline, cpu_baseline and roofline are in hand, and runs every leg under its
watchdog; each leg returns a detail dict (written to stderr / --detail), bench.compact_line() keeps one short summary per leg.

  two_stream_leg            config 2 as two contiguous shards on two HIP streams
  reference_semantics_leg   the CLASS DEFAULTS a drop-in caller of the reference's loop gets (fresh tensors per step)
  strong_scaling_leg        65 536 envs IN TOTAL over the world (eager / hipGraph K = 8, 32), or one rank's shard emulated here
  device_guard_check        N > 1: fe_env_device == LOCAL_RANK; a step issued with another device current (TSE:28, 42)
  fused_rollout_legs        SURVEY 8f.2: K steps per launch with an in-kernel linear / MLP / LSTM policy
"""
from __future__ import annotations

import gc
import statistics
import time

import torch

from bench import (CONFIGS, HBM_PEAK_GBPS, SETTLE_MS, Dist, KernelTrain, auto_repeats, hbm_bytes,  # noqa: E402  (bench.py registers itself
                   make_series)                                                                    # as `bench` before importing this)

def two_stream_leg(args, steps: int):
    """The headline workload as TWO contiguous shards (rank 0 / 1 of 2: the same envs, the evaluation env in the second)
    stepped on two HIP streams.  Envs are independent, so a rollout loop that evaluates its policy per shard (a
    double-buffered sampler, examples/double_buffered_rollout.py) lets one shard's start-up chain and launch boundary
    overlap the other shard's store stream across steps.  Launches go through the C ABI (pre-generated actions, as in the
    headline's kernel-interval loop).  Never part of `value`."""
    import finenvs_amd

    name, N, A, W = CONFIGS[2]
    prices, day_id, _ = make_series(A)
    dev = "cuda:0"
    parts, streams = [], [torch.cuda.Stream(), torch.cuda.Stream()]
    for r in range(2):
        env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, rank=r, world_size=2,
                                        redraw="device", seed=1234, obs_buffers=2)
        n = env.num_envs
        g = torch.Generator(device=dev).manual_seed(7 + r)
        acts = [(torch.rand((n, A), generator=g, device=dev) * 2 - 1).float() for _ in range(8)]
        env.reset()
        parts.append((env, acts, torch.empty((n,), dtype=torch.float64, device=dev), torch.empty((n,), dtype=torch.int32, device=dev)))
    k2 = max(min(max(steps, 20), 400), 200)  # (a 27 us step: 200 launch pairs ~ 5 ms per run)
    times = []
    for rep in range(4):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s_ in streams:
            s_.wait_event(e0)
        for k in range(k2 if rep else 4 * k2):  # the discarded first run also carries the GPU past its post-idle clock transient
            for (env, acts, rew, done), s_ in zip(parts, streams):
                rc = env._step_fn(env._handle_v, acts[k % 8].data_ptr(), env._obs_ring[k % 2].data_ptr(), rew.data_ptr(), done.data_ptr(),
                                  s_.cuda_stream)
        for s_ in streams:
            torch.cuda.current_stream().wait_stream(s_)
        e1.record()
        torch.cuda.synchronize()
        from finenvs_amd import _lib as _fl

        _fl.check(rc)
        if rep:
            times.append(e0.elapsed_time(e1) / k2)
    ms = statistics.median(times)
    Bh = hbm_bytes(W, A, 8)
    del parts
    torch.cuda.empty_cache()
    return {"workload": name, "what": "two contiguous shards (rank 0 / 1 of 2) of the same 65 536 envs on two HIP streams, C-ABI launches, "
                                      "HIP events around the whole loop", "envs": N, "ms_per_step_all_envs": ms,
            "value": N / ms * 1e3, "unit": "env-steps/s", "achieved_GBps": Bh * N / (ms * 1e-3) / 1e9,
            "frac_of_8TBps": Bh * N / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "launches_per_run": 2 * k2}


def reference_semantics_leg(args, steps: int, repeats: int = 0):
    """What a drop-in caller of the reference's loop gets (examples/time_series/PPO_LSTM_training_SPY.py:22-30): the CLASS
    DEFAULTS -- fresh observation / reward / done tensors per step (obs_buffers=0), redraw='torch' with the reference's
    per-step host read of the evaluation env's done flag -- on the headline workload (config 2), timed with the headline's
    block protocol (R blocks of exactly `steps` steps between synchronising fences after SETTLE_MS of untimed steps, median
    block).  Never part of `value`."""
    import finenvs_amd

    name, N, A, W = CONFIGS[2]
    prices, day_id, _ = make_series(A)
    torch.manual_seed(1234)
    env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, device_id=0)
    assert env.obs_buffers == 0 and env.redraw == "torch" and env.obs_dtype == torch.float64
    g = torch.Generator(device="cuda:0").manual_seed(7)
    actions = [(torch.rand((N, A), generator=g, device="cuda:0") * 2 - 1).float() for _ in range(8)]
    states = env.reset()
    for i in range(20):
        states, _, _, _ = env.step(actions[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(8):
        states, _, _, _ = env.step(actions[i % 8])
    torch.cuda.synchronize()
    est = (time.perf_counter() - t0) / 8
    R = auto_repeats(repeats, steps, est)
    for i in range(int(SETTLE_MS * 1e-3 / est)):  # the clock transient after the idle period of construction (run_workload.settle)
        states, _, _, _ = env.step(actions[i % 8])
    blocks = []
    for _ in range(R):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            states, rew, done, _ = env.step(actions[i % 8])
        torch.cuda.synchronize()
        blocks.append(time.perf_counter() - t0)
    med = statistics.median(blocks)
    del env, states
    torch.cuda.empty_cache()
    return {"workload": name, "what": "class defaults: obs_buffers=0 (fresh tensors per step), redraw='torch' (per-step host read, TSE:510), "
                                      "f64 observations, eager env.step(actions) loop; the headline's block protocol",
            "value": N * steps / med, "unit": "env-steps/s", "ms_per_step": med / steps * 1e3,
            "ms_per_step_min": min(blocks) / steps * 1e3, "ms_per_step_max": max(blocks) / steps * 1e3, "steps": steps,
            "blocks": R}


STRONG_TOTAL_ENVS = 65536  # BASELINE.json's metric: "env-steps/sec at 64k envs, 1/2/4/8 MI355X" read as ONE 64k-env job


def strong_scaling_leg(args, D: Dist, steps: int, warmup: int, world: int = None, rank: int = None):
    """The STRONG-scaling reading of the metric: 65 536 envs IN TOTAL, sharded contiguously over the world (8 GPUs: 8 192 envs
    per GPU ~ 5 us of HBM time per step -- launch-bound, where the fused / graphed forms earn their keep).  Two launch modes:
    `eager` (env.step per step, --redraw's mode, trajectory slots written by the kernel) and `graph_k8` / `graph_k32` (rollout.GraphedRollout,
    8 / 32 steps per hipGraph replay, redraw='device' as capture requires); with N > 1 each with and without the trajectory
    all-gather (one packed chunk per `steps` eager steps resp. per 8-step replay, asynchronous, double-buffered).
    `world` / `rank`: emulate one rank's shard of a larger world on THIS GPU without collectives (the N = 1 run's preview of
    the per-GPU step time at 2 / 4 / 8 GPUs: the measured basis of DESIGN.md section 7's strong-scaling rows)."""
    import finenvs_amd
    from finenvs_amd.rollout import GraphedRollout
    from finenvs_amd.trajectory import TrajectoryBuffer

    emulated = world is not None
    w = world if emulated else D.world
    r = rank if emulated else D.rank
    dev = D.dev
    _, _, A, W = CONFIGS[2]
    prices, day_id, _ = make_series(A)
    gathering = D.multi and not emulated
    out = {"total_envs": STRONG_TOTAL_ENVS, "world": w, "emulated_on_one_gpu": emulated}

    def fence(trajs=()):
        for t in trajs:
            t.drain()
        if D.dist is not None and not emulated:
            if D.backend != "nccl":
                torch.cuda.synchronize()
            D.barrier()
        torch.cuda.synchronize()

    def blocks_of(run_block, n_steps, trajs=(), r_blocks=None):
        run_block()
        fence(trajs)
        t0 = time.perf_counter()
        run_block()
        fence(trajs)
        est = (time.perf_counter() - t0) / n_steps
        est = est if emulated else D.max_over_ranks(est)
        R = r_blocks or auto_repeats(args.repeats, n_steps, est)
        for _ in range(int(min(4000, SETTLE_MS * 1e-3 / max(est, 1e-9)) / n_steps) + 1):  # settle (run_workload.settle)
            run_block()
        ts = []
        for _ in range(R):
            fence(trajs)
            t0 = time.perf_counter()
            run_block()
            fence(trajs)
            dt = time.perf_counter() - t0
            ts.append(dt if emulated else D.max_over_ranks(dt))
        med = statistics.median(ts)
        return {"value": STRONG_TOTAL_ENVS * n_steps / med, "us_per_step": med / n_steps * 1e6, "us_per_step_min": min(ts) / n_steps * 1e6,
                "us_per_step_max": max(ts) / n_steps * 1e6, "blocks": R, "steps_per_block": n_steps}

    # ---- eager
    env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=STRONG_TOTAL_ENVS, rank=r, world_size=w,
                                    device_id=D.local_rank, redraw=args.redraw, seed=1234, obs_buffers=2)
    n = env.num_envs
    out["envs_per_gpu"] = n
    g = torch.Generator(device=dev).manual_seed(7 + r)
    actions = [(torch.rand((n, A), generator=g, device=dev) * 2 - 1).float() for _ in range(8)]
    traj = TrajectoryBuffer(steps, n, A, device=dev)
    env.reset()
    gather = [False]

    def eager_block():
        for i in range(steps):
            if traj.full():
                if gather[0]:
                    traj.all_gather_async(defer=True)
                else:
                    traj.clear()
            a, rw, d = traj.next_slot()
            env.step(actions[i % 8], rewards_out=rw, dones_out=d, actions_out=a)
            if i == 2:
                traj.issue_deferred()

    for _ in range(max(1, warmup // max(steps, 1))):
        eager_block()
    out["eager"] = {"no_all_gather": blocks_of(eager_block, steps, (traj,))}
    kt = KernelTrain(env, actions)
    out["eager"]["kernel_us"] = statistics.median(kt.run(400) for _ in range(3)) * 1e3
    out["eager"]["launch"] = env.launch_info()
    if gathering:
        gather[0] = True
        traj.clear()
        eager_block()
        out["eager"]["with_all_gather"] = blocks_of(eager_block, steps, (traj,))
        gather[0] = False
        fence((traj,))
        out["eager"]["packed_bytes_per_rank_per_chunk"] = traj._nbytes
    del env, traj, kt
    # ---- hipGraph, K steps per replay (two graphs over two trajectory chunks: chunk i is gathered while graph 1 - i replays).
    # K = 8 is the leg VERDICT round 4 named; K = 32 shows what a longer replay buys once a collective per replay is in the loop
    # (its host start + latency are per replay, the steps per replay amortise them)
    for K in (8, 32):
        steps_g = (steps + 2 * K - 1) // (2 * K) * (2 * K)  # whole replays, both graphs equally often
        env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=STRONG_TOTAL_ENVS, rank=r, world_size=w,
                                        device_id=D.local_rank, redraw="device", seed=1234, obs_buffers=2)
        trajs = [TrajectoryBuffer(K, n, A, device=dev) for _ in range(2)]
        rolls = [GraphedRollout(env, lambda obs, k: actions[k % 8], K, trajectory=t) for t in trajs]
        pending = [None, None]
        gathered = [None, None]

        def graph_block():
            for j in range(steps_g // K):
                i = j & 1
                if pending[i] is not None:
                    pending[i].wait()  # (stream-level for RCCL) chunk i has left before graph i overwrites it
                    pending[i] = None
                rolls[i].run()
                if gather[0]:
                    if gathered[i] is None:
                        gathered[i] = torch.empty((D.dist.get_world_size(), trajs[i]._nbytes), dtype=torch.uint8, device=dev)
                    pending[i] = D.dist.all_gather_into_tensor(gathered[i].view(-1), trajs[i]._packed, async_op=True)

        def graph_fence_extra():
            for i in (0, 1):
                if pending[i] is not None:
                    pending[i].wait()
                    pending[i] = None

        class _Drain:  # fence() drains these like a TrajectoryBuffer
            drain = staticmethod(graph_fence_extra)

        key = f"graph_k{K}"
        out[key] = {"no_all_gather": blocks_of(graph_block, steps_g, (_Drain,))}
        if gathering:
            gather[0] = True
            out[key]["with_all_gather"] = blocks_of(graph_block, steps_g, (_Drain,))
            gather[0] = False
            fence((_Drain,))
            out[key]["packed_bytes_per_rank_per_chunk"] = trajs[0]._nbytes
        del rolls, trajs, env, gathered
    gc.collect()
    torch.cuda.empty_cache()
    return out


def device_guard_check(D: Dist):
    """N > 1 only: the paths a one-GPU box cannot reach.  (i) fe_env_device(env) == LOCAL_RANK on every rank; (ii) a step() issued
    while ANOTHER device is current (the reference's `device_id` argument works without torch.cuda.set_device, TSE:28, 42;
    csrc/fe_env.hip DeviceGuard) launches on the env's device, returns tensors there, leaves the caller's current device
    unchanged, and computes what a step with the env's device current computes.  Returns a pass / fail record (never raises)."""
    import finenvs_amd

    rec = {"local_rank": D.local_rank, "devices_visible": torch.cuda.device_count()}
    try:
        _, _, A, W = CONFIGS[1]
        prices, day_id, _ = make_series(A)
        mk = lambda: finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=1024, evaluate=True,  # noqa: E731
                                               device_id=D.local_rank)
        env, twin = mk(), mk()
        rec["env_device"] = int(env._lib.fe_env_device(env._handle))
        ok = rec["env_device"] == D.local_rank
        a = (torch.rand((1024, A), device=D.dev) * 2 - 1).float()
        o0, r0, d0, _ = twin.step(a)
        if rec["devices_visible"] > 1:
            other = (D.local_rank + 1) % rec["devices_visible"]
            with torch.cuda.device(other):
                before = torch.cuda.current_device()
                o1, r1, d1, _ = env.step(a)
                after = torch.cuda.current_device()
            rec.update(other_device=other, current_device_before=before, current_device_after=after,
                       outputs_on=str(o1.device), current_device_restored=torch.cuda.current_device() == D.local_rank)
            torch.cuda.synchronize(D.dev)
            same = bool(torch.equal(o0, o1) and torch.equal(r0, r1) and torch.equal(d0, d1))
            rec["same_result_as_with_own_device_current"] = same
            ok = ok and before == other and after == other and str(o1.device) == D.dev and same and rec["current_device_restored"]
        else:
            rec["other_device"] = None
            rec["note"] = "one device visible: the cross-device half needs the multi-GPU node"
        rec["pass"] = bool(ok)
    except Exception as exc:  # noqa: BLE001
        rec["pass"] = False
        rec["error"] = f"{type(exc).__name__}: {exc}"
    return rec


def fused_rollout_legs(args):
    """SURVEY 8f.2 legs, reported beside the headline (never part of `value`): K env steps per launch with the policy
    evaluated in the kernel, at the headline's 65 536 envs x 1 asset.  The observation is never written to HBM, so these
    are not HBM-roofline numbers: the MLP / LSTM legs report the f32 MFMA rate of their contractions instead."""
    import finenvs_amd
    from finenvs_amd.rollout import FusedLinearRollout, FusedLSTMRollout, FusedMLPRollout

    _, N, A, _ = CONFIGS[2]
    prices, day_id, _ = make_series(A)
    g = torch.Generator().manual_seed(0)
    legs = []
    for form, W, K in (("linear_table", 64, 32), ("mlp_h64", 64, 32), ("lstm_h128", 4, 8), ("lstm_h1024", 4, 2)):
        try:
            env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device",  # (in-kernel redraws)
                                            seed=1234, obs_buffers=1)
            flop = 0.0
            if form == "linear_table":
                roll = FusedLinearRollout(env, torch.randn((W, 5), dtype=torch.float64, generator=g) * 2, 0.0, form="table")
                policy = "clamp(<window, weights (W, 5)>), log-return part precomputed as an indicator table"
            elif form == "mlp_h64":
                H = 64
                roll = FusedMLPRollout(env, torch.randn((5 * W, H), generator=g) * (8.0 / W ** 0.5), torch.randn(H, generator=g) * 0.3,
                                       torch.randn(H, generator=g) / H ** 0.5, 0.0)
                flop = 2.0 * N * A * (4 * W) * H
                policy = "Linear(5W, 64) -> ELU -> Linear(64, 1), first layer on v_mfma_f32_32x32x2_f32"
            else:
                H = int(form.split("_h")[1])
                torch.manual_seed(0)
                lstm, lin = torch.nn.LSTM(5, H, batch_first=True), torch.nn.Linear(H, 1)
                with torch.no_grad():
                    lstm.weight_ih_l0[:, :4].mul_(6.0 * H ** 0.5)
                roll = FusedLSTMRollout.from_modules(env, lstm, lin)
                flop = 2.0 * N * A * 4 * H * (8 * W + H * (W - 1))
                policy = (f"the reference's actor: LSTM(5, {H}) over the W rows -> Linear({H}, 1) -> tanh, gates on v_mfma_f32_32x32x2_f32"
                          + (", recurrent weights streamed from L2 (the reference example's hidden_dim)" if H > 128 else ", recurrent weights in registers"))
            roll.run(K, record_actions=True)
            times = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                roll.run(K, record_actions=True)
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1) / K)
            ms = statistics.median(times)
            leg = {"form": form, "policy": policy, "envs": N, "num_assets": A, "window": W, "steps_per_launch": K,
                   "us_per_step": round(ms * 1e3, 2), "value": round(N / ms * 1e3, 1), "unit": "env-steps/s"}
            if flop:  # MFMA-bound legs: the contraction's FLOPs as executed against the dense f32 MFMA peak (256 CUs x 256 FLOP/cycle x 2.4 GHz)
                leg["mfma_f32_tflops"] = round(flop / ms / 1e9, 1)
                leg["mfma_f32_peak_tflops"] = 157.3
                leg["roofline"] = {"bound": "mfma", "achieved": round(flop / ms / 1e9, 1), "peak": 157.3, "unit": "TFLOP/s",
                                   "frac": round(flop / ms / 1e9 / 157.3, 3), "traffic": None,
                                   "note": "launch time from HIP events around K-step launches (policy + accounting); the f32-input MFMA "
                                           "shares the vector ALUs with the activations (SQ_VALU_MFMA_COEXEC_CYCLES = 0, profiles/r02g_lstm_summary.md)"}
            legs.append(leg)
            del env, roll
            torch.cuda.empty_cache()
        except Exception as exc:  # noqa: BLE001
            legs.append({"form": form, "error": f"{type(exc).__name__}: {exc}"})
    return legs
