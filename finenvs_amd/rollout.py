"""K rollout steps as ONE hipGraph launch.

The reference's rollout loop (examples/time_series/PPO_LSTM_training_SPY.py:22-30) pays
Python + launch overhead per step: agent.step -> env.step -> agent.store.  At 64k envs the fused
env kernel takes ~35 us, the same order as that overhead.  ``GraphedRollout`` captures K
iterations of  policy(obs) -> env.step(actions) -> trajectory.store(...)  into a single
torch.cuda.CUDAGraph (a hipGraph on ROCm); ``run()`` replays it with one launch call.

Requirements: env.redraw == "device" and training or evaluate-free stepping (nothing on the
captured path may synchronise with the host), and K a multiple of env.obs_buffers so that the
observation a replay ends on is the buffer the next replay starts from.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

from .trajectory import TrajectoryBuffer


class GraphedRollout:
    def __init__(self, env, policy: Callable[[torch.Tensor, int], torch.Tensor], num_steps: int,
                 trajectory: Optional[TrajectoryBuffer] = None, warmup: int = 2):
        if env.evaluate:
            raise ValueError("evaluate mode reads a device counter on the host every step; it cannot be captured")
        if env.redraw != "device":
            raise ValueError('GraphedRollout needs redraw="device" (redraw="torch" syncs on the eval env\'s done flag)')
        if env.obs_buffers < 1 or num_steps % env.obs_buffers != 0:
            raise ValueError("num_steps must be a multiple of env.obs_buffers (>= 1)")
        if trajectory is not None and trajectory.T != num_steps:
            raise ValueError("trajectory buffer must hold exactly num_steps steps")
        self.env, self.policy, self.K, self.traj = env, policy, int(num_steps), trajectory
        self.rewards: List[torch.Tensor] = []
        self.dones: List[torch.Tensor] = []
        dev = env._dev
        # warm up on a side stream (allocator + lazy init), as torch.cuda.graph requires
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            self.obs = env.reset()
            for _ in range(warmup):
                self._iterate(record=False)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._iterate(record=True)
        # self.obs now names the static buffer holding the newest observation after each replay

    def _iterate(self, record: bool) -> None:
        env, traj = self.env, self.traj
        if traj is not None:
            traj.clear()
        obs = self.obs
        rews, dones = [], []
        for k in range(self.K):
            actions = self.policy(obs, k)
            obs, rew, done, _ = env.step(actions)
            if traj is not None:
                traj.store(actions, rew, done)
            rews.append(rew)
            dones.append(done)
        self.obs = obs
        if record:
            self.rewards, self.dones = rews, dones

    def run(self) -> torch.Tensor:
        """Replay the K captured steps; returns the newest observation (a static buffer)."""
        self.graph.replay()
        if self.traj is not None:
            self.traj.t = self.K
        return self.obs
