"""Device-resident trajectory buffer + RCCL all-gather.

Replaces the compact-field half of the reference's PPO buffer
(finenvs/agents/PPO/buffer.py): ``store`` is one slot write into preallocated
time-major tensors instead of a ``torch.cat`` per step (buffer.py:33-56, O(T^2)
bytes moved), and ``returns_and_advantages`` is one reverse-scan kernel per env
instead of a Python loop over T (buffer.py:80-100), with the reference's dtype
discipline (f32 discount factor, f64 carry, f32 results).

Multi-GPU: envs are sharded contiguously, one process per GPU; the only
exchange is ``all_gather`` of these compact fields -- actions (T, n, A) f32,
rewards (T, n) f64, dones (T, n) i32 -- once per T-step chunk, as ONE collective
over a single packed byte buffer (RCCL over xGMI when the process group is
"nccl").  Observations are never gathered: they stay sharded with their learner.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


class TrajectoryBuffer:
    def __init__(self, num_steps: int, num_envs: int, num_assets: int = 1, device: str = "cuda:0",
                 host_rehearsal: bool = False, capacity: Optional[int] = None):
        """``capacity`` (>= num_envs, same on every rank) sizes the env axis of the buffers, so that ranks
        owning shards of different sizes (N not divisible by the world size) can still exchange chunks with
        one all-gather; ``join_shards`` drops the padding again."""
        self.T, self.N, self.A = int(num_steps), int(num_envs), int(num_assets)
        self.C = int(capacity) if capacity is not None else self.N
        if self.C < self.N:
            raise ValueError("capacity must be >= num_envs")
        self.device = torch.device(device)
        if self.device.type != "cuda" and not host_rehearsal:
            raise RuntimeError("TrajectoryBuffer lives in HBM; host tensors are accepted only with host_rehearsal=True, "
                               "which exists to rehearse the all-gather plumbing over gloo (no kernels run there)")
        T, N, A = self.T, self.C, self.A  # the env axis is laid out with `capacity` slots
        # one allocation per chunk, three typed views: [rewards f64 | actions f32 | dones i32];
        # two chunks so that a chunk can be in flight on the collective stream while the next fills
        self._nbytes = T * N * 8 + T * N * A * 4 + T * N * 4
        self._chunks = [torch.zeros((self._nbytes,), dtype=torch.uint8, device=self.device) for _ in range(2)]
        self._views = [self._typed(c) for c in self._chunks]
        # per-slot views, built once (tensor indexing costs microseconds of host time per call)
        n = self.N
        self._slots = [[(v[0][t, :n], v[1][t, :n], v[2][t, :n]) for t in range(T)] for v in self._views]
        self._pending = [None, None]   # outstanding collective per chunk
        self._gathered = [None, None]  # its output buffer
        self._cur = 0
        self.t = 0
        self._native = self.device.type == "cuda"
        self._lib = _lib.load() if self._native else None

    def _typed(self, packed: torch.Tensor, lead: Tuple[int, ...] = ()):
        T, N, A = self.T, self.C, self.A
        o1 = T * N * 8
        o2 = o1 + T * N * A * 4
        flat = packed.reshape(-1, self._nbytes) if lead else packed
        if lead:
            G = flat.shape[0]
            return (flat[:, o1:o2].contiguous().view(torch.float32).view(G, T, N, A),
                    flat[:, :o1].contiguous().view(torch.float64).view(G, T, N),
                    flat[:, o2:].contiguous().view(torch.int32).view(G, T, N))
        return (packed[o1:o2].view(torch.float32).view(T, N, A), packed[:o1].view(torch.float64).view(T, N),
                packed[o2:].view(torch.int32).view(T, N))

    # the chunk being filled
    @property
    def actions(self) -> torch.Tensor:
        return self._views[self._cur][0][:, : self.N]

    @property
    def rewards(self) -> torch.Tensor:
        return self._views[self._cur][1][:, : self.N]

    @property
    def dones(self) -> torch.Tensor:
        return self._views[self._cur][2][:, : self.N]

    @property
    def _packed(self) -> torch.Tensor:
        return self._chunks[self._cur]

    def _stream(self) -> int:
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        if _raw_stream is not None:
            return _raw_stream(idx)
        return torch.cuda.current_stream(self.device).cuda_stream

    def __len__(self) -> int:
        return self.t

    def full(self) -> bool:
        return self.t >= self.T

    def clear(self) -> None:
        self.t = 0

    def store(self, actions: torch.Tensor, rewards: torch.Tensor, dones: torch.Tensor) -> None:
        if self.t >= self.T:
            raise IndexError("trajectory buffer is full; call clear()")
        if self._native:
            if actions.dtype is not torch.float32:
                actions = actions.float()
            st = self._stream()
            sa, sr, sd = self._slots[self._cur][self.t]  # slot t starts at row t of the capacity-strided buffers
            _lib.check(self._lib.fe_traj_store(
                0, self.N, self.A, actions.contiguous().data_ptr(), rewards.data_ptr(), dones.data_ptr(),
                sa.data_ptr(), sr.data_ptr(), sd.data_ptr(), st))
        else:  # host tensors: only the gloo rehearsal of the collective uses this
            self.actions[self.t].copy_(actions.reshape(self.N, self.A))
            self.rewards[self.t].copy_(rewards)
            self.dones[self.t].copy_(dones)
        self.t += 1

    def next_slot(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Zero-copy form of ``store``: views of slot t -- actions (N, A), rewards (N,), dones (N,) --
        for the policy and ``env.step(actions, rewards_out=..., dones_out=...)`` to write into
        directly; advances t."""
        if self.t >= self.T:
            raise IndexError("trajectory buffer is full; call clear()")
        t = self.t
        self.t += 1
        return self._slots[self._cur][t]

    def returns_and_advantages(self, values: torch.Tensor, last_values: torch.Tensor, gamma: float = 0.99
                               ) -> Tuple[torch.Tensor, torch.Tensor]:
        """(returns, advantages), both (T, N) f32, per buffer.py:80-100.  values (T, N) f32,
        last_values (N,) f32."""
        if not self._native:
            raise RuntimeError("returns_and_advantages runs on the GPU only (no CPU path)")
        T, N = self.t, self.N
        values = values.reshape(T, N).float().contiguous()
        last_values = last_values.reshape(N).float().contiguous()
        ret = torch.empty((T, N), dtype=torch.float32, device=self.device)
        adv = torch.empty((T, N), dtype=torch.float32, device=self.device)
        st = self._stream()
        rew, don = self.rewards[:T], self.dones[:T]
        if self.C != self.N:  # the scan kernel wants dense (T, N) inputs
            rew, don = rew.contiguous(), don.contiguous()
        _lib.check(self._lib.fe_traj_returns(rew.data_ptr(), don.data_ptr(), values.data_ptr(),
                                             last_values.data_ptr(), T, N, float(gamma), ret.data_ptr(),
                                             adv.data_ptr(), st))
        return ret, adv

    # ------------------------------------------------------------------ multi-GPU exchange
    def all_gather(self, group=None, out: Optional[torch.Tensor] = None):
        """One blocking collective: every rank receives every rank's packed chunk.

        Returns ``(actions (G, T, n, A), rewards (G, T, n), dones (G, T, n), packed)``
        (all ranks must own the same n)."""
        import torch.distributed as dist

        G = dist.get_world_size(group)
        if out is None:
            out = torch.empty((G, self._nbytes), dtype=torch.uint8, device=self.device)
        dist.all_gather_into_tensor(out.view(-1), self._packed, group=group)
        a, r, d = self._typed(out, lead=(G,))
        return a, r, d, out

    def all_gather_async(self, group=None) -> None:
        """Start gathering the chunk just filled on the collective's own stream and switch to the
        other chunk, so the exchange over xGMI overlaps the next T env steps.  ``wait_gathered``
        returns the result; a chunk is waited for automatically before it is refilled."""
        import torch.distributed as dist

        G = dist.get_world_size(group)
        i = self._cur
        if self._gathered[i] is None:
            self._gathered[i] = torch.empty((G, self._nbytes), dtype=torch.uint8, device=self.device)
        self._pending[i] = dist.all_gather_into_tensor(self._gathered[i].view(-1), self._chunks[i], group=group,
                                                       async_op=True)
        self._cur = 1 - i
        self._wait(self._cur)  # the chunk about to be refilled must have left
        self.t = 0

    def _wait(self, i: int) -> None:
        if self._pending[i] is not None:
            self._pending[i].wait()  # stream-level wait for NCCL/RCCL; blocks the host only for gloo
            self._pending[i] = None

    def wait_gathered(self):
        """(actions, rewards, dones) of the most recently started gather, as (G, T, n, ...) tensors."""
        i = 1 - self._cur
        self._wait(i)
        if self._gathered[i] is None:
            raise RuntimeError("no gather has been started")
        return self._typed(self._gathered[i], lead=(self._gathered[i].shape[0],))

    def drain(self) -> None:
        for i in (0, 1):
            self._wait(i)

    @staticmethod
    def join_shards(x: torch.Tensor, total_envs: int) -> torch.Tensor:
        """(G, T, capacity, ...) gathered field -> (T, total_envs, ...): shards concatenated in rank order,
        capacity padding dropped (shards as produced by ``shard_range``)."""
        from .environments.time_series_env import shard_range

        G = x.shape[0]
        parts = []
        for r in range(G):
            lo, hi = shard_range(total_envs, r, G)
            parts.append(x[r][:, : hi - lo])
        return torch.cat(parts, dim=1)
