#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for W in 1 6 8; do
  FE_ROLLOUT_WAVES=$W python3 -m finenvs_amd.csrc.build --force > /dev/null 2>&1
  for EB in 16 32 64; do
    FE_TILE_ENVS=$EB python3 - <<PY 2>&1 | grep -v amdgpu
import os, sys, torch
sys.path.insert(0, ".")
import finenvs_amd
from finenvs_amd.rollout import FusedLinearRollout
from bench import CONFIGS, make_series
name, N, A, W = CONFIGS[2]
prices, day_id, _ = make_series(A)
env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", obs_buffers=1)
roll = FusedLinearRollout(env, torch.randn((W, 5), dtype=torch.float64) * 2, 0.0)
K = 32
roll.run(K)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(5): roll.run(K)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / (5 * K)
print(f"waves=$W EB=$EB: {ms*1e3:8.2f} us/step  {N/ms/1e6:8.2f} G env-steps/s")
PY
  done
done
python3 -m finenvs_amd.csrc.build --force > /dev/null 2>&1
