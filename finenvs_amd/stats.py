"""Episode statistics kept on the device.

The reference's agents track, per rollout step and with a host sync each time
(finenvs/agents/PPO/PPO_agent.py:120-132, 146-163): a running return per env, the list of
returns of finished *training* episodes (all envs but the last), and the return of the last
finished *evaluation* episode (the last env); at log time they report
``len(list), mean(list), std(list)``.  ``EpisodeStats`` binds three small device buffers to
the env; the fused step kernel updates them -- the finished-episode sums as PER-ENV partials
(N, 3) f64, each slot written only by the lane that owns the env -- and ``read()`` adds the
partials up in a fixed order (``fe_env_stats_reduce``: one workgroup, lane-strided sums and a
halving tree; the test oracle restates it) and fetches the numbers with a single
device-to-host copy at log time.  Mean / std are therefore bit-stable from run to run and do
not depend on the kernel's tile walk, launch geometry or form (the reference reduces its list
in a fixed order too, PPO_agent.py:146-163).
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from . import _lib


class EpisodeStats:
    def __init__(self, env):
        if env.evaluate:
            raise ValueError("EpisodeStats follows the training-mode bookkeeping of the reference's agents")
        self.env = env
        dev = env._dev
        self.running_returns = torch.zeros((env.num_envs,), dtype=torch.float32, device=dev)
        self._acc = torch.zeros((env.num_envs, 3), dtype=torch.float64, device=dev)  # per-env partial sums
        self._sums = torch.zeros((3,), dtype=torch.float64, device=dev)
        self._eval = torch.zeros((2,), dtype=torch.float32, device=dev)
        _lib.check(env._lib.fe_env_bind_stats(env._handle, self.running_returns.data_ptr(), self._acc.data_ptr(),
                                              self._eval.data_ptr()))
        self._epoch = env._binding_epoch

    def _check_bound(self) -> None:
        if self.env._binding_epoch != self._epoch:
            raise RuntimeError("the env was resized (env_indices assigned with another length) after this EpisodeStats was bound: "
                               "create a new one")

    def close(self) -> None:
        if self.env._binding_epoch != self._epoch:
            return  # the env object these buffers were bound to is gone
        _lib.check(self.env._lib.fe_env_bind_stats(self.env._handle, None, None, None))

    def read(self, reset: bool = True) -> Dict[str, float]:
        """One D2H copy: what PPOAgent.log_progress prints (PPO_agent.py:146-163)."""
        self._check_bound()
        _lib.check(self.env._lib.fe_env_stats_reduce(self.env._handle, self._sums.data_ptr(), self.env._stream()))
        acc = self._sums.cpu()
        ev = self._eval.cpu()
        n, s, ss = float(acc[0]), float(acc[1]), float(acc[2])
        mean = s / n if n > 0 else float("nan")
        var = (ss - n * mean * mean) / (n - 1) if n > 1 else float("nan")  # unbiased, as torch.std
        out = {
            "num_training_episodes": int(n),
            "mean_training_return": mean,
            "std_dev_training_return": math.sqrt(max(var, 0.0)) if n > 1 else float("nan"),
            "evaluation_return": float(ev[0]) if ev[1] > 0 else None,
            "num_evaluation_episodes": int(ev[1]),
        }
        if reset:
            self._acc.zero_()
            self._eval.zero_()
        return out
