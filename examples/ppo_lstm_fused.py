#!/usr/bin/env python3
"""PPO with the reference's LSTM actor on the fused rollout: the loop of
/root/reference/examples/time_series/PPO_LSTM_training_SPY.py:22-30 with the K env steps between two ``agent.train``
calls done by ONE kernel launch (actor evaluated in the kernel, actions sampled from the caller's noise, agent.store's
fields written into a trajectory chunk whose ``states`` are 16-byte descriptors).  The learner is plain PyTorch-ROCm
user code (agents are out of this repo's scope): clipped-surrogate PPO with an LSTM actor and an LSTM critic, shaped
like finenvs/agents/PPO/{PPO_agent,continuous_actor,critic}.py.

    python examples/ppo_lstm_fused.py [--envs 4096] [--steps 16] [--iters 5] [--hidden 64] [--window 4]

What runs where:
    rollout   : FusedLSTMRollout.run(K, noise, std, trajectory)       one launch per K steps, nothing written but
                                                                       actions / rewards / dones / descriptors
    values    : FusedLSTMRollout(..., "none").forward(descriptors)    all (K + 1) x N states in one launch, no observations
    returns   : TrajectoryBuffer.returns_and_advantages               one reverse-scan kernel (buffer.py:80-100)
    update    : torch autograd on minibatches rendered from descriptors (PPO_agent.py:175-196)
"""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from finenvs_amd import TimeSeriesEnv  # noqa: E402
from finenvs_amd.data import synthetic  # noqa: E402
from finenvs_amd.rollout import FusedLSTMRollout  # noqa: E402
from finenvs_amd.stats import EpisodeStats  # noqa: E402
from finenvs_amd.trajectory import TrajectoryBuffer  # noqa: E402


class LSTMHead(torch.nn.Module):
    """LSTM(5, H) over the window, Linear(H, 1) on the last hidden state (the shape of the reference's LSTMNetwork)."""

    def __init__(self, hidden: int, squash: bool):
        super().__init__()
        self.lstm = torch.nn.LSTM(5, hidden, num_layers=1, batch_first=True)
        self.last = torch.nn.Linear(hidden, 1)
        self.squash = squash

    def forward(self, states: torch.Tensor) -> torch.Tensor:
        out = self.last(self.lstm(states)[0][:, -1, :])
        return torch.tanh(out) if self.squash else out


def main(envs=4096, steps=16, iters=5, hidden=64, window=4, epochs=2, minibatches=4, seed=0, quiet=False):
    torch.manual_seed(seed)
    dev = "cuda:0"
    prices, day_id, _ = synthetic.synthetic_series(12, 1, 390, 1234)
    env = TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=window, num_envs=envs, redraw="device", seed=seed,
                        obs_dtype=torch.float32)  # the learner consumes states.float() (PPO_agent.py:101)
    actor, critic = LSTMHead(hidden, True).to(dev), LSTMHead(hidden, False).to(dev)
    log_std = torch.nn.Parameter(torch.full((1,), math.log(0.5), device=dev))
    opt_a = torch.optim.Adam(list(actor.parameters()) + [log_std], 3e-4)
    opt_c = torch.optim.Adam(critic.parameters(), 3e-4)
    traj = TrajectoryBuffer(steps, envs, 1, states=True)
    stats = EpisodeStats(env)
    roll = FusedLSTMRollout.from_modules(env, actor.lstm, actor.last)
    value_head = FusedLSTMRollout.from_modules(env, critic.lstm, critic.last, output_activation="none")
    gen = torch.Generator(device=dev).manual_seed(seed)
    clip, ent_coef, gamma = 0.2, 0.01, 0.99
    history = []
    t0 = time.perf_counter()
    for it in range(iters):
        # ---- rollout: K env steps, one launch (agent.step + env.step + agent.store, K times) ----
        std = float(log_std.detach().exp())
        noise = torch.randn((steps, envs, 1), generator=gen, device=dev)
        actions, rewards, dones = roll.run(steps, noise=noise, std=std, record_means=True, trajectory=traj)
        with torch.no_grad():
            old_logp = torch.distributions.Normal(roll.means, std).log_prob(actions)           # (K, N, 1)
            # ---- values of the K stored states and of the bootstrap state: the critic on their descriptors ----
            values = value_head.forward(traj.obs_src, traj.obs_pos).reshape(steps + 1, envs)          # (K+1, N)
            returns, advantages = traj.returns_and_advantages(values[:steps], values[steps], gamma)     # (K, N) f32
        # ---- update: minibatches rendered from the descriptors (sample = env * steps + step, buffer.py:102-109) ----
        total = envs * steps
        flat = lambda x: x.reshape(steps, envs).t().reshape(total)  # (K, N[,1]) -> env-major samples
        f_act, f_logp, f_adv, f_ret = flat(actions), flat(old_logp), flat(advantages), flat(returns)
        for _ in range(epochs):
            perm = torch.randperm(total, device=dev)
            for mb in perm.chunk(minibatches):
                states = traj.minibatch_states(env, mb)  # (B, W, 5) f32, rendered now
                dist = torch.distributions.Normal(actor(states).squeeze(-1), log_std.exp())
                ratio = (dist.log_prob(f_act[mb]) - f_logp[mb]).exp()
                adv = f_adv[mb]
                surrogate = torch.minimum(ratio * adv, ratio.clamp(1 - clip, 1 + clip) * adv).mean()
                loss_a = -(surrogate + ent_coef * dist.entropy().mean())
                opt_a.zero_grad()
                loss_a.backward()
                opt_a.step()
                loss_c = ((f_ret[mb] - critic(states).squeeze(-1)) ** 2).mean()
                opt_c.zero_grad()
                loss_c.backward()
                opt_c.step()
        for head, net in ((roll, actor), (value_head, critic)):  # the updated networks go back into the kernels
            head.set_weights(net.lstm.weight_ih_l0, net.lstm.weight_hh_l0, net.lstm.bias_ih_l0, net.lstm.bias_hh_l0,
                             net.last.weight, float(net.last.bias.detach()))
        traj.clear()
        log = stats.read(reset=True)
        history.append((float(loss_c.detach()), float(rewards.mean()), log))
        if not quiet:
            print(f"iter {it}: critic loss {history[-1][0]:.4f}  mean step reward {history[-1][1]:+.5f}  std {std:.3f}  "
                  f"finished episodes {log['num_training_episodes']}", flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if not quiet:
        print(f"{iters} iterations of {steps} steps x {envs} envs in {dt:.2f} s ({iters * steps * envs / dt / 1e6:.2f} M env-steps/s "
              f"including the learner)")
    return history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--hidden", type=int, default=64, choices=[32, 64, 128, 256, 512, 1024])
    ap.add_argument("--window", type=int, default=4)
    a = ap.parse_args()
    main(a.envs, a.steps, a.iters, a.hidden, a.window)
