"""Sharding parity on the GPU box: two ranks (both on cuda:0, gloo for the exchange) each own a
contiguous shard of the envs; their gathered trajectories must equal a single-process env over all
N envs, bit for bit -- env n -> day n mod D, eval env on the last rank only, device redraw there."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_TOTAL, A, W, T = 1001, 2, 8, 70  # odd: the two shards differ in size (501 / 500), buffers padded to 501


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _series():
    from finenvs_amd.data import synthetic

    return synthetic.synthetic_series(5, A, 30, 21)


def _actions(t):
    g = torch.Generator().manual_seed(1000 + t)
    return (torch.rand((N_TOTAL, A), generator=g) * 2 - 1).float()


def _worker(rank, world, port, q):
    import torch.distributed as dist

    import finenvs_amd
    from finenvs_amd.trajectory import TrajectoryBuffer

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        prices, day_id, _ = _series()
        env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N_TOTAL, rank=rank,
                                        world_size=world, redraw="device", seed=77)
        lo, hi = finenvs_amd.shard_range(N_TOTAL, rank, world)
        assert env.num_envs == hi - lo and env.env_offset == lo and env.global_num_envs == N_TOTAL
        buf = TrajectoryBuffer(T, env.num_envs, A, device=env.device, capacity=(N_TOTAL + world - 1) // world)
        env.reset()
        for t in range(T):
            a, r, d = buf.next_slot()
            a.copy_(_actions(t)[lo:hi].to(env.device))
            env.step(a, rewards_out=r, dones_out=d)
        buf.all_gather_async()
        acts, rews, dones = buf.wait_gathered()
        torch.cuda.synchronize()
        rews, dones = TrajectoryBuffer.join_shards(rews, N_TOTAL), TrajectoryBuffer.join_shards(dones, N_TOTAL)
        q.put((rank, rews.cpu().numpy(), dones.cpu().numpy(), acts.cpu().numpy(), env.env_indices.cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_two_sharded_ranks_equal_one_unsharded_env():
    import torch.multiprocessing as mp

    import finenvs_amd

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=240) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # the single-process reference run
    prices, day_id, _ = _series()
    env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N_TOTAL, redraw="device", seed=77)
    rew = np.empty((T, N_TOTAL)); done = np.empty((T, N_TOTAL), dtype=np.int32)
    env.reset()
    for t in range(T):
        _, r, d, _ = env.step(_actions(t).to(env.device))
        rew[t], done[t] = r.cpu().numpy(), d.cpu().numpy()
    assert done.sum() >= 2 * N_TOTAL  # two episode ends: the eval env redrew its day twice
    for rank, rews, dones, acts, idx in got:
        # every rank holds every shard after the gather: (G, T, n) -> (T, N)
        assert np.array_equal(rews, rew)
        assert np.array_equal(dones, done)
        lo, hi = finenvs_amd.shard_range(N_TOTAL, rank, world)
        assert np.array_equal(idx, env.env_indices.cpu().numpy()[lo:hi])  # incl. the eval env's redrawn day
        assert np.array_equal(acts[rank][3][: hi - lo], _actions(3)[lo:hi].numpy())


def test_double_buffered_rollout_equals_the_unsharded_graphed_rollout():
    """DoubleBufferedRollout: two shards, two hipGraphs, two streams -- bit for bit what ONE GraphedRollout over the
    unsharded env gives (per-env policy), over several replays that cross day ends, including the evaluation env's
    device redraws (it lives in the second shard)."""
    import finenvs_amd as fe
    from finenvs_amd.data import synthetic
    from finenvs_amd.rollout import DoubleBufferedRollout, GraphedRollout

    prices, day_id, _ = synthetic.synthetic_series(6, 1, 24, 5, 0.05)
    N, W, K = 1001, 8, 6
    kw = dict(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=11, obs_buffers=2)
    w = torch.linspace(-3.0, 3.0, 5, dtype=torch.float64, device="cuda")

    def policy(obs, k):  # per env, deterministic: a function of the env's own last row only
        return torch.tanh((obs[:, -1, :] * w).sum(dim=1, keepdim=True) * (k + 1)).float()

    whole = fe.TimeSeriesEnv(**kw)
    ref = GraphedRollout(whole, policy, K)
    dbl = DoubleBufferedRollout(lambda r, ws: fe.TimeSeriesEnv(rank=r, world_size=ws, **kw), policy, K, shards=2)
    assert dbl.num_envs == N and [e.num_envs for e in dbl.envs] == [501, 500]
    redraws = 0
    for rep in range(12):
        o_ref = ref.run()
        dbl.run()
        o_dbl = dbl.join()
        torch.cuda.synchronize()
        assert torch.equal(torch.cat(o_dbl), o_ref), f"replay {rep}: observations"
        assert torch.equal(dbl.joined_rewards(), torch.stack(ref.rewards)), f"replay {rep}: rewards"
        assert torch.equal(dbl.joined_dones(), torch.stack(ref.dones)), f"replay {rep}: dones"
        assert torch.equal(torch.cat([e.cash for e in dbl.envs]), whole.cash)
        assert torch.equal(torch.cat([e.env_indices for e in dbl.envs]), whole.env_indices)
        redraws += int(torch.stack(ref.dones)[:, -1].sum())
    assert redraws >= 2  # the evaluation env finished (and redrew its day identically) more than once
    with pytest.raises(ValueError):
        DoubleBufferedRollout(lambda r, ws: fe.TimeSeriesEnv(**kw), policy, K)  # make_env ignored rank / world_size
