"""GPU box: throughput of the fused in-kernel-policy rollout (fe_env_rollout_linear) per config."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd
from finenvs_amd.rollout import FusedLinearRollout
from bench import CONFIGS, make_series
for cfg in [int(x) for x in (sys.argv[1:] or ["2"])]:
    name, N, A, W = CONFIGS[cfg]
    prices, day_id, _ = make_series(A)
    for eb in (None, 4, 8, 16, 32, 64, 128, 256):
        if eb: os.environ["FE_TILE_ENVS"] = str(eb)
        else: os.environ.pop("FE_TILE_ENVS", None)
        env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", obs_buffers=1)
        w = torch.randn((W, 5), dtype=torch.float64) * 2
        roll = FusedLinearRollout(env, w, 0.0, form=os.environ.get('FUSED_FORM', 'window'))
        K = int(os.environ.get('FUSED_K', '32'))
        roll.run(K, record_actions=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        reps = 5
        for _ in range(reps): roll.run(K, record_actions=True)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (reps * K)
        print(f"cfg {cfg} EB={env.launch_info()['tile_envs']:4d}: {ms*1e3:9.2f} us/step  {N/ms/1e6*1e3/1e3:8.2f} G env-steps/s", flush=True)
        del env, roll
        if A > 1 and eb and eb * A >= 256: break
