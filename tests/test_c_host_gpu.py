"""The C ABI used from plain C (examples/c_host/fe_c_demo.c: HIP runtime only, no Python, no torch)
must give the same numbers as the same run through the Python binding."""
import math
import os
import re
import shutil
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_plain_c_host_matches_python_binding(tmp_path):
    import finenvs_amd

    exe = str(tmp_path / "fe_c_demo")
    libdir = os.path.join(REPO, "finenvs_amd", "csrc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["gcc", "-O2", os.path.join(REPO, "examples", "c_host", "fe_c_demo.c"), "-I",
                           os.path.join(REPO, "include"), "-I", f"{rocm}/include", "-L", libdir, "-lfinenvs_amd",
                           "-L", f"{rocm}/lib", "-lamdhip64", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{rocm}/lib", "-lm",
                           "-o", exe])
    N, W, steps = 3000, 16, 120
    out = subprocess.check_output([exe, str(N), str(W), str(steps)], text=True, timeout=120)
    got = {k: float(v) for k, v in re.findall(r"(\w+)=([-+0-9.eE]+)", out.splitlines()[-1])}
    # the same run through fe_env_step_notify (host flag polled from C instead of reading dones[-1]): identical output
    out_notify = subprocess.check_output([exe, str(N), str(W), str(steps), "notify"], text=True, timeout=120)
    assert out_notify.splitlines()[-1] == out.splitlines()[-1]
    assert got["eval_dones"] >= 1

    # float64 actions through fe_env_step_promoted from C (two steps of three; the f32 steps between them stay promoted)
    out_promoted = subprocess.check_output([exe, str(N), str(W), str(steps), "promoted"], text=True, timeout=120)
    got_p = {k: float(v) for k, v in re.findall(r"(\w+)=([-+0-9.eE]+)", out_promoted.splitlines()[-1])}

    # the same series, state and actions through the Python binding
    days, bars = 6, 50
    T = days * bars
    series = np.empty((T, 4))
    px = 100.0
    for t in range(T):
        o = px * (1.0 + 0.0007 * math.sin(0.37 * t))
        c = o * (1.0 + 0.0009 * math.cos(0.11 * t))
        series[t] = (o, max(o, c) * 1.0004, min(o, c) * 0.9996, c)
        px = c
    day_id = np.repeat(np.arange(days), bars)
    from finenvs_amd.stats import EpisodeStats

    for mode, want in (("plain", got), ("promoted", got_p)):
        env = finenvs_amd.TimeSeriesEnv(prices=series, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=42)
        D = env.price_environments.shape[0]
        env.env_indices.copy_(torch.arange(N, device=env.device) % D)  # the demo starts the eval env on day (N-1) % D
        env._counters[1] = 0
        stats = EpisodeStats(env)
        env.reset()
        rew_sum, dones = 0.0, 0
        n1 = np.arange(1, N + 1, dtype=np.float64)
        for s in range(steps):
            a64 = np.sin(0.013 * n1 * (s + 1))
            a = torch.from_numpy(a64 if (mode == "promoted" and s % 3 != 2) else a64.astype(np.float32)).reshape(N, 1).to(env.device)
            obs, rew, done, _ = env.step(a)
            r = rew.cpu().numpy()
            for x in r:  # same summation order as the C loop
                rew_sum += float(x)
            dones += int(done.sum())
        assert env.shares_promoted == (mode == "promoted")
        cash_sum = 0.0
        for x in env.cash.cpu().numpy().reshape(-1):
            cash_sum += float(x)
        obs_sum = 0.0
        for x in obs[-1].cpu().numpy().reshape(-1):
            obs_sum += float(x)
        assert want["dones"] == dones and dones > 0, mode
        assert want["cash_sum"] == cash_sum, mode
        # sin/cos come from two libms (C vs Python's): identical on this image, but allow an ulp-level drift
        assert want["reward_sum"] == pytest.approx(rew_sum, rel=1e-12), mode
        assert want["last_obs_sum"] == pytest.approx(obs_sum, rel=1e-12), mode
        # the fused episode statistics and their fixed-order reduction, read from C and through the binding: the same bits
        stats.read(reset=False)
        sums = stats._sums.cpu().numpy()
        assert want["stat_episodes"] == sums[0] > 0 and want["stat_sum"] == sums[1] and want["stat_sumsq"] == sums[2], mode
        stats.close()
    assert got_p["reward_sum"] != got["reward_sum"]  # the promoted arithmetic is a different (the reference's f64) arithmetic
