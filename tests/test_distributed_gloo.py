"""World-size-2 and -8 rehearsals of the multi-GPU path on CPU (gloo): contiguous env shards, eval env on the
last rank, one packed all-gather of the compact trajectory fields per chunk.  (On MI355X the same code
runs with backend "nccl" = RCCL; bench.py --gpus N exercises it.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from finenvs_amd.environments.time_series_env import shard_range
from finenvs_amd.trajectory import TrajectoryBuffer

T, A = 5, 3
# (world, total envs): 15 envs over 2 ranks -> shards of 8 and 7 (capacity 8); 27 envs over 8 ranks -> three
# shards of 4 and five of 3 (capacity 4), N not divisible by the world size, eval env on rank 7
CASES = [(2, 15), (8, 27)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fill(buf, lo, n):
    """Deterministic per-(step, global env) values so any rank can predict any shard."""
    for t in range(T):
        gidx = torch.arange(lo, lo + n, dtype=torch.float64)
        actions = (gidx.unsqueeze(1) * 10 + torch.arange(A) + 1000 * t).float()
        rewards = gidx * 0.5 - t
        dones = ((gidx.long() + t) % 3 == 0).int()
        buf.store(actions, rewards, dones)


def _worker(rank, world, port, q, N_TOTAL):
    CAP = (N_TOTAL + world - 1) // world
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(N_TOTAL, rank, world)
        n = hi - lo
        buf = TrajectoryBuffer(T, n, A, device="cpu", host_rehearsal=True, capacity=CAP)
        _fill(buf, lo, n)
        assert buf.full()
        actions, rewards, dones, _ = buf.all_gather()
        ok = True
        for r in range(world):
            l2, h2 = shard_range(N_TOTAL, r, world)
            ref = TrajectoryBuffer(T, h2 - l2, A, device="cpu", host_rehearsal=True, capacity=CAP)
            _fill(ref, l2, h2 - l2)
            m = h2 - l2
            ok &= torch.equal(actions[r][:, :m], ref.actions) and torch.equal(rewards[r][:, :m], ref.rewards) and torch.equal(dones[r][:, :m], ref.dones)
        # join_shards drops the padding: (T, N_TOTAL) in global env order
        joined = TrajectoryBuffer.join_shards(rewards, N_TOTAL)
        gidx = torch.arange(N_TOTAL, dtype=torch.float64)
        ok &= joined.shape == (T, N_TOTAL) and torch.equal(joined[2], gidx * 0.5 - 2)
        # the overlapped form used by bench.py: start, keep filling the other chunk, then collect
        buf.all_gather_async()
        assert len(buf) == 0
        _fill(buf, lo + 1000, n)          # the next chunk fills while the first is in flight
        a2, r2, d2 = buf.wait_gathered()
        for r in range(world):
            l2, h2 = shard_range(N_TOTAL, r, world)
            ref = TrajectoryBuffer(T, h2 - l2, A, device="cpu", host_rehearsal=True, capacity=CAP)
            _fill(ref, l2, h2 - l2)
            m = h2 - l2
            ok &= torch.equal(a2[r][:, :m], ref.actions) and torch.equal(r2[r][:, :m], ref.rewards) and torch.equal(d2[r][:, :m], ref.dones)
        buf.all_gather_async()
        a3, r3, d3 = buf.wait_gathered()
        ok &= float(r3[rank][0, 0]) == (lo + 1000) * 0.5
        ok &= tuple(a3.shape) == (world, T, CAP, A)
        buf.drain()
        # the deferred form of bench.py's loop: chunks switch at once, the collective starts a couple of steps later
        # (issue_deferred) -- or, if nobody calls that, with the next wait / drain / gather
        _fill(buf, lo + 2000, n)
        buf.all_gather_async(defer=True)
        ok &= len(buf) == 0 and buf._deferred is not None and buf._pending == [None, None]
        a_, r_, d_ = buf.next_slot()      # the new chunk is already being filled
        a_.zero_(); r_.zero_(); d_.zero_()
        buf.issue_deferred()
        ok &= buf._deferred is None
        a4, r4, d4 = buf.wait_gathered()
        ok &= all(float(r4[rr][0, 0]) == (shard_range(N_TOTAL, rr, world)[0] + 2000) * 0.5 for rr in range(world))
        buf.clear()
        _fill(buf, lo + 3000, n)
        buf.all_gather_async(defer=True)  # never issued explicitly: the wait does it
        a5, r5, d5 = buf.wait_gathered()
        ok &= float(r5[rank][0, 0]) == (lo + 3000) * 0.5
        buf.clear()
        _fill(buf, lo + 4000, n)
        buf.all_gather_async(defer=True)  # ... and so does drain()
        buf.drain()
        ok &= buf._deferred is None and buf._pending == [None, None]
        # ... and the BLOCKING gather: a chunk still deferred leaves before it (program order of collectives on every rank)
        buf.clear()
        _fill(buf, lo + 5000, n)
        buf.all_gather_async(defer=True)
        _fill(buf, lo + 6000, n)
        a6, r6, d6, _ = buf.all_gather()      # the chunk being filled now
        ok &= buf._deferred is None and buf._pending[1 - buf._cur] is not None
        ok &= all(float(r6[rr][0, 0]) == (shard_range(N_TOTAL, rr, world)[0] + 6000) * 0.5 for rr in range(world))
        a7, r7, d7 = buf.wait_gathered()      # ... and the deferred one arrived too
        ok &= all(float(r7[rr][0, 0]) == (shard_range(N_TOTAL, rr, world)[0] + 5000) * 0.5 for rr in range(world))
        buf.drain()
        # the eval env (last global env) is owned by the last rank only
        owns_eval = hi == N_TOTAL
        ok &= owns_eval == (rank == world - 1)
        # a step-time barrier + max-over-ranks reduction like bench.py's
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok &= float(t) == world
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", CASES)
def test_trajectory_all_gather_gloo(world, n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, n_total)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(r, True) for r in range(world)]


def test_shards_cover_days_like_the_single_process_env():
    """env n -> day n mod D must not depend on how many ranks there are."""
    D, N = 6, 50
    single = np.arange(N) % D
    for world in (2, 3, 8):
        parts = []
        for r in range(world):
            lo, hi = shard_range(N, r, world)
            parts.append(np.arange(lo, hi) % D)
        assert np.array_equal(np.concatenate(parts), single)


def _state_of(step, gidx):
    """Deterministic descriptors per (row, global env): row t = the state at step t, row T = the bootstrap state."""
    src = (gidx.long() * 100 + step * 4)
    pos = (gidx.unsqueeze(1) * 0.25 + torch.arange(A) + step).double()
    return src, pos


def _states_worker(rank, world, port, q, N_TOTAL):
    CAP = (N_TOTAL + world - 1) // world
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(N_TOTAL, rank, world)
        n = hi - lo
        gidx = torch.arange(lo, lo + n, dtype=torch.float64)
        buf = TrajectoryBuffer(T, n, A, device="cpu", host_rehearsal=True, capacity=CAP, states=True)
        ok = True
        for chunk in range(2):
            if chunk == 0:
                buf.begin(_state_of(0, gidx))
            for t in range(T):
                a, r, d = buf.next_slot()
                a.fill_(float(t)); r.fill_(float(chunk)); d.zero_()
                src, pos = buf.state_slot()  # what env.step(descriptors_out=...) would fill
                s2, p2 = _state_of(chunk * T + t + 1, gidx)
                src.copy_(s2); pos.copy_(p2)
            buf.all_gather_async()
            a_g, r_g, d_g, src_g, pos_g = buf.wait_gathered(with_states=True)
            ok &= tuple(src_g.shape) == (world, T + 1, CAP) and tuple(pos_g.shape) == (world, T + 1, CAP, A)
            for rr in range(world):
                l2, h2 = shard_range(N_TOTAL, rr, world)
                g2 = torch.arange(l2, h2, dtype=torch.float64)
                for row in range(T + 1):  # every rank holds every rank's states, rows 0..T of this chunk
                    s2, p2 = _state_of(chunk * T + row, g2)
                    ok &= torch.equal(src_g[rr, row, : h2 - l2], s2) and torch.equal(pos_g[rr, row, : h2 - l2], p2)
            # the chunk now being filled starts from the previous chunk's bootstrap row
            s0, p0 = _state_of((chunk + 1) * T, gidx)
            ok &= torch.equal(buf.obs_src[0], s0) and torch.equal(buf.obs_pos[0], p0)
        joined = TrajectoryBuffer.join_shards(src_g, N_TOTAL)
        ok &= tuple(joined.shape) == (T + 1, N_TOTAL)
        # the blocking form carries the states too (the chunk being filled: only its carried row 0 is set so far)
        # geometry handshake before anyone renders a foreign descriptor: equal on all ranks passes, one odd rank raises
        # on EVERY rank (it is a collective, so nobody is left waiting)
        class _Env:
            _dev = torch.device("cpu")

            def __init__(self, W):
                self.W = W

            def geometry(self):
                return torch.tensor([64, 518, self.W, A], dtype=torch.int64)

        TrajectoryBuffer.check_geometry(_Env(128))
        try:
            TrajectoryBuffer.check_geometry(_Env(128 if rank else 64))
            ok = False
        except ValueError as exc:
            ok &= "geometry" in str(exc)
        a_b, r_b, d_b, src_b, pos_b, packed_b = buf.all_gather(with_states=True)
        ok &= tuple(src_b.shape) == (world, T + 1, CAP) and tuple(packed_b.shape) == (world, buf._nbytes)
        ok &= torch.equal(src_b[rank, 0, :n], buf.obs_src[0])
        buf.drain()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_state_descriptors_travel_with_the_all_gather_gloo():
    """states=True: the gathered chunk carries every rank's state descriptors (rows 0..T), and a chunk that follows
    another starts from its bootstrap row."""
    world, n_total = 2, 15
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_states_worker, args=(r, world, port, q, n_total)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(r, True) for r in range(world)]
