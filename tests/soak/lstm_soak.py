"""GPU box: a longer random sweep of the LSTM rollout against the oracle loop than the test suite runs (all six hidden
sizes, 1..30 sleeves, W 1..9, partial tiles, both modes, sampled / mean actions, trajectory output, forward()).

    python tests/soak/lstm_soak.py [cases] [seed]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import finenvs_amd as fe  # noqa: E402
from finenvs_amd.rollout import FusedLSTMRollout  # noqa: E402
from finenvs_amd.trajectory import TrajectoryBuffer  # noqa: E402
from oracle import fe_oracle as fo  # noqa: E402
from tests.test_lstm_rollout_gpu import _make, _modules, _packed, t2n  # noqa: E402

fo.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)


def same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.shape != b.shape or a.tobytes() != b.tobytes():
        bad = np.argwhere(a != b)
        raise SystemExit(f"MISMATCH {what}: {len(bad)} elements, first at {bad[0].tolist() if len(bad) else '?'}")


for c in range(cases):
    H = int(rng.choice([32, 64, 128, 256, 512, 1024], p=[0.25, 0.2, 0.2, 0.15, 0.12, 0.08]))
    A = int(rng.choice([1, 1, 2, 3, 5, 7, 12, 30]))
    W = int(rng.integers(1, 10))
    budget = 400 if H <= 128 else (160 if H <= 512 else 60)  # pairs: the oracle's cost grows with H^2
    N = int(rng.integers(1, max(2, budget // A + 1)))
    evaluate, sample, use_traj = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    ref, env = _make(fe, fo, N, A, W, 5, 30, 0.05, evaluate, seed=int(rng.integers(1, 1000)))
    lstm, lin = _modules(H, seed=int(rng.integers(1, 1000)), gain=6.0 if H <= 128 else 2.0)
    whh, wx, wout, bout = _packed(fo, lstm, lin)
    roll = FusedLSTMRollout.from_modules(env, lstm, lin)
    if H > 128:
        roll.split = bool(rng.integers(0, 2))  # the per-time-step path or the fused large-H kernel
    obs = ref.reset().copy()
    g = torch.Generator(device="cuda").manual_seed(c)
    std = np.float32(0.4)
    what = f"case {c} H={H} A={A} W={W} N={N} eval={evaluate} sample={sample} traj={use_traj} split={roll.split}"
    for rep in range(3):
        K = int(rng.integers(1, 6))
        traj = TrajectoryBuffer(K, N, A, states=True) if use_traj else None
        noise = torch.randn((K, N, A), generator=g, device="cuda") if sample else None
        acts, rews, dones = roll.run(K, noise=noise, std=float(std) if sample else None, trajectory=traj)
        for k in range(K):
            if traj is not None:
                same(t2n(traj.states(env, k)), obs, what + f" rep {rep} stored state {k}")
            a_ref = fo.policy_lstm(obs, whh, wx, wout, bout)
            if rep == 0 and k == 0:  # forward() on the same descriptors gives the same means
                src, pos = (traj.obs_src[0], traj.obs_pos[0]) if traj is not None else (None, None)
                if src is not None:
                    same(t2n(roll.forward(src, pos)), a_ref, what + " forward()")
            if sample:
                smp = np.clip((a_ref + (std * t2n(noise[k])).astype(np.float32)).astype(np.float32), np.float32(-1), np.float32(1))
                if not evaluate:
                    smp[N - 1] = a_ref[N - 1]
                a_ref = smp
            obs, r_ref, d_ref, _ = ref.step(a_ref)
            obs = obs.copy()
            same(t2n(acts[k]), a_ref, what + f" rep {rep} step {k} actions")
            same(t2n(rews[k]), r_ref, what + f" rep {rep} step {k} rewards")
            same(t2n(dones[k]), d_ref, what + f" rep {rep} step {k} dones")
        same(t2n(env.cash), ref.cash, what + " cash")
        same(t2n(env.margin), ref.margin, what + " margin")
        same(t2n(env.env_indices), ref.env_idx, what + " env_idx")
        if evaluate and int(ref.n_terminated[0]) == N:
            env.reset_evaluation_metrics()
            ref.terminated[:] = 0; ref.episode_returns[:] = 0; ref.n_terminated[0] = 0
    print(what, "ok", flush=True)
print(f"{cases} cases ok")
