"""GPU box: interleaved A/B of step-kernel builds in ONE process (same box, same clocks).

    python tools/ab_step.py <config> name[=tile,grid] [name[=tile,grid] ...]

`name` is "product" or an experiment build made by finenvs_amd.csrc.build.build_variant (loaded by explicit
path from finenvs_amd/csrc/variants/); "=tile,grid" overrides the launch geometry (0 = automatic).
Every arm first replays 24 steps and must equal the product bit for bit (observation, rewards, dones, state);
then R rounds of K back-to-back launches per arm, arms interleaved, one HIP-event pair per block.
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd  # noqa: E402
from bench import CONFIGS, make_series  # noqa: E402
from finenvs_amd import _lib  # noqa: E402

cfg = int(sys.argv[1])
arms = sys.argv[2:] or ["product"]
name, N, A, W = CONFIGS[cfg]
prices, day_id, _ = make_series(A)
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(7)
actions = [(torch.rand((N, A), generator=g, device=dev) * 2 - 1).float() for _ in range(8)]
F32 = os.environ.get("AB_F32") == "1"  # f32 observations
OBS_DT = torch.float32 if F32 else torch.float64
obs_bytes = N * W * 5 * A * (4 if F32 else 8)
nbuf = 2 if 2 * obs_bytes < 200e9 else 1
VAR = os.path.join(os.path.dirname(_lib.LIB_PATH), "variants")


def make(arm):
    nm, _, geo = arm.partition("=")
    lib = _lib.load() if nm == "product" else _lib.load(os.path.join(VAR, f"libfinenvs_amd.{nm}.so"))
    env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device",
                                    seed=1234, obs_buffers=nbuf, obs_dtype=OBS_DT, _native=lib)
    if geo:
        t, gr = (int(x) for x in geo.split(","))
        env.set_launch(t, gr)
    return env


def replay(env, steps=24):
    env.reset()
    out = None
    for i in range(steps):
        out = env.step(actions[i % 8])
    torch.cuda.synchronize()
    obs = out[0] if obs_bytes < 30e9 else out[0][::509]  # a sample of envs where the observation is huge
    return [obs.clone(), out[1].clone(), out[2].clone(), env.cash.clone(), env.margin.clone(),
            env.long_shares.clone(), env.short_shares.clone(), env._spot0.clone(), env.env_indices.clone()]


BIG = obs_bytes * nbuf > 100e9  # only one env of this size fits at a time: arms run one after the other
ref = None if BIG else replay(make("product"))  # (BIG: the first arm's own replay is the reference -- list "product" first)
K = 100 if obs_bytes < 1e9 else 10
R = 15 if obs_bytes < 1e9 else 5
stream = torch.cuda.current_stream().cuda_stream
times = {a: [] for a in arms}


SHARED_RING = None  # ONE observation ring for every arm: HBM write bandwidth depends on where a buffer lies
                    # (measured: the same build 3.21 vs 3.63 ms at config 3 on two rings), so arms must share it


def block(env):
    global SHARED_RING
    if SHARED_RING is None and not BIG:
        SHARED_RING = [torch.empty((N, W, 5 * A), dtype=OBS_DT, device=dev) for _ in range(nbuf)]
    # (BIG: every arm writes its own ring, one env at a time -- nothing of an arm may stay referenced once it is deleted)
    obs_b = [t.data_ptr() for t in (env._obs_ring if BIG else SHARED_RING)]
    rew = torch.empty((N,), dtype=torch.float64, device=dev)
    done = torch.empty((N,), dtype=torch.int32, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    act_b = torch.empty((N, A), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    e0.record()
    if os.environ.get("AB_FULL") == "1":  # the full form of the kernel (trajectory action copy), as bench.py's loop launches it
        for i in range(K):
            env._lib.fe_env_step_traj(env._handle_v, actions[i % 8].data_ptr(), obs_b[i % nbuf], rew.data_ptr(), done.data_ptr(),
                                      act_b.data_ptr(), None, None, stream)
    else:
        for i in range(K):
            env._step_fn(env._handle_v, actions[i % 8].data_ptr(), obs_b[i % nbuf], rew.data_ptr(), done.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3


def check(arm, e):
    global ref
    got = replay(e)
    if ref is None:
        ref = got
    ok = all(torch.equal(a, b) or (a.dtype.is_floating_point and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)))
                                      for a, b in zip(ref, got))
    print(f"{arm:28s} launch {e.launch_info()}  parity vs product: {'OK' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        sys.exit(1)


if BIG:
    for arm in arms:
        e = make(arm)
        check(arm, e)
        block(e)
        times[arm] = [block(e) for _ in range(R)]
        del e
        torch.cuda.empty_cache()
else:
    envs = {}
    for arm in arms:
        envs[arm] = make(arm)
        check(arm, envs[arm])
        envs[arm]._obs_ring = []  # the timed blocks write the shared ring
        torch.cuda.empty_cache()
    for r in range(R + 1):
        for arm, env in envs.items():
            t = block(env)
            if r > 0:  # round 0 warms up
                times[arm].append(t)
base = statistics.median(times[arms[0]])
for arm in arms:
    t = times[arm]
    med = statistics.median(t)
    print(f"{arm:28s} us/step median {med:9.2f}  min {min(t):9.2f}  max {max(t):9.2f}   vs {arms[0]}: {med / base:6.3f}   "
          f"obs-write {obs_bytes / med / 1e6:6.2f} TB/s", flush=True)
